// ss_kernels.h -- launch wrappers of the gfx950 kernels (ss_kernels.hip), called from ss_api.hip.
// Templated on the reference's Real type R (float / double); explicit instantiations live in ss_kernels.hip.
#pragma once
#include "ss_device.h"
#include "ss_prims.h"

template <class R>
void ss_launch_aabb(const R* d_xyz, uint32_t n, R* d_partial, R* d_out6, SSMailSlot mail, hipStream_t st);
// d_partial holds SS_AABB_PARTIAL_WORDS values of R (<= 1024 blocks, SS_AABB_PARTIAL_STRIDE each); the mail's VALUE is 1 if a coordinate is not finite
#define SS_AABB_PARTIAL_STRIDE 7
#define SS_AABB_PARTIAL_WORDS (1024 * SS_AABB_PARTIAL_STRIDE)
template <class R>
void ss_launch_inside_flags(const R* d_xyz, uint32_t n, const R amin[3], const R amax[3], uint8_t* f8, uint32_t* f32, hipStream_t st);
template <class R>
void ss_launch_compact_xyz(const R* d_xyz, uint32_t n, const uint32_t* f32, const uint32_t* offs, R* out, hipStream_t st);
template <class R>
void ss_launch_cell_keys(const SSDevT<R>& P, const R* d_xyz, uint32_t* keys, uint32_t* vals, hipStream_t st);
template <class R>
void ss_launch_emit_copies(const SSDevT<R>& P, const R* xyz, const uint32_t* copy_offset, const uint32_t* occ_rank, uint32_t* keys, uint32_t* vals, hipStream_t st);
template <class R>
void ss_launch_density_sub(const SSDevT<R>& P, uint32_t n_copies, const ss_pos<R>* cpos, const uint32_t* cidx, const uint32_t* ckey, const uint32_t* cell_start, const uint32_t* occ_sub, R* rho, int mode, uint32_t* nb_count, const unsigned long long* nb_ptr, uint32_t* nb_idx, bool fast_div, const uint32_t* owned_list, const uint32_t* n_owned_dev, uint32_t n_owned_bound, hipStream_t st);
template <class R>
void ss_launch_make_posvol(const SSDevT<R>& P, const ss_pos<R>* pos_sorted, const uint32_t* perm, const R* rho, ss_real4<R>* posvol, ss_real4<R>* posvol_by_index, hipStream_t st);
template <class R>
void ss_launch_mark_blocks(const SSDevT<R>& P, const uint32_t* cell_start, uint32_t ncells, uint32_t* block_flag, hipStream_t st);
void ss_launch_verify_fast_div(float h, float rh, uint32_t* bad, hipStream_t st);
template <class R>
void ss_launch_splat_bounds(const SSDevT<R>& P, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* counts, uint32_t* bound, hipStream_t st);
template <class R>
void ss_launch_splat_gather(const SSDevT<R>& P, const ss_real4<R>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const unsigned long long* tile_off, ss_real4<R>* arena, uint32_t* arena_idx, uint32_t* counts, uint32_t* large_flag, hipStream_t st);
template <class R>
void ss_launch_splat_gather_large(const SSDevT<R>& P, const ss_real4<R>* posvol, const ss_real4<R>* posvol_by_index, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, const uint32_t* large_list, const uint32_t* n_large_dev, const uint32_t* counts, const unsigned long long* tile_off, ss_real4<R>* arena, uint32_t* arena_idx, hipStream_t st);
template <class R>
void ss_launch_splat_row_table(const SSDevT<R>& P, uint2* tab, hipStream_t st);
template <class R>
void ss_launch_splat_fused(const SSDevT<R>& P, const ss_real4<R>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint2* row_tab, const uint32_t* active_xyz, uint32_t n_active, R* G, ss_real2<R>* blk_minmax, uint32_t* trunc, bool full_levelset, const uint32_t* list, const uint32_t* n_list_dev, const uint32_t* redo_mask, unsigned long long* facebits, uint32_t* counts, uint32_t* big, hipStream_t st);
template <class R>
void ss_launch_splat_accumulate_big(const SSDevT<R>& P, const ss_real4<R>* arena, const uint32_t* arena_idx, const unsigned long long* tile_off, const uint32_t* counts, const uint32_t* active_xyz, R* G, ss_real2<R>* blk_minmax, uint32_t* trunc, bool full_levelset, bool second_pass, bool exact_first, const uint32_t* redo_mask, unsigned long long* facebits, const uint32_t* big, uint32_t* err, hipStream_t st);
void ss_launch_splat_certify_big(const SSDevT<float>& P, const ss_real4<float>* posvol, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* block_slot, uint32_t* counts, ss_real2<float>* blk_minmax, uint32_t* trunc, unsigned long long* facebits, uint32_t* need_mask, hipStream_t st);
template <class R>
void ss_launch_select_redo(const SSDevT<R>& P, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* block_slot, const uint32_t* trunc, const unsigned long long* facebits, uint32_t* redo_mask, const uint32_t* counts, unsigned long long* stats, uint32_t* big, hipStream_t st);
void ss_launch_publish_stats(const unsigned long long* stats, const uint32_t* n_redo, const uint32_t* n_large, const uint32_t* err, SSMailSlot m0, SSMailSlot m1, SSMailSlot m2, SSMailSlot m3, hipStream_t st);
#define SS_MC_REC 28  // words of a marching-cubes block's record in mc_nb (k_mc_neighbours)
template <class R>
void ss_launch_mc_neighbours(const SSDevT<R>& P, const uint32_t* mc_xyz, uint32_t n_mc, const uint32_t* block_slot, const uint32_t* mc_slot, const uint32_t* certified, uint32_t* mc_nb, hipStream_t st);
template <class R>
void ss_launch_mc_count(const SSDevT<R>& P, const R* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, uint32_t n_mc, unsigned long long* masks, uint32_t* vcount, uint32_t* tcount, hipStream_t st);
template <class R>
void ss_launch_mc_emit(const SSDevT<R>& P, const R* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, const uint32_t* mc_slot, uint32_t n_mc, const unsigned long long* masks, const uint32_t* vbase, const uint32_t* tbase, R* vertices, unsigned long long* vkeys, uint32_t* triangles, hipStream_t st);
void ss_launch_stream_probe(bool copy, const void* in, void* out, size_t n_float4, float* sink, hipStream_t st);
void ss_launch_widen(const uint32_t* in, size_t n, unsigned long long* out, hipStream_t st);
template <class R>
void ss_launch_levelset_box(const SSDevT<R>& P, const R* G, const uint32_t* block_slot, const int lo[3], const int ext[3], R* out, hipStream_t st);

// fused scans of the host flow (one dispatch each, ss_prims.h)
template <class R>
void ss_launch_sorted_gather_runs(const SSDevT<R>& P, uint32_t n, const R* xyz, const uint32_t* perm, ss_pos<R>* pos_sorted, const uint32_t* sorted_keys, uint32_t ncells, uint32_t* first, const uint32_t* occ_sub, uint8_t* owned, hipStream_t st);
void ss_launch_cell_table_scan(const uint32_t* first, uint32_t ncells, uint32_t* cell_start, uint32_t* state, hipStream_t st);
template <class R>
void ss_launch_classify_scan(const SSDevT<R>& P, const R* xyz, uint32_t* copy_offset, uint32_t* sub_flag, uint32_t* state, SSMailSlot mail, hipStream_t st);
void ss_launch_flag_scan(const uint32_t* flag, uint32_t n, uint32_t* rank, uint32_t* list, uint32_t* total_dev, uint32_t* state, SSMailSlot mail, hipStream_t st);
void ss_launch_owned_scan(uint32_t n_copies, const uint8_t* owned, uint32_t* own_list, uint32_t* n_owned_dev, uint32_t* state, hipStream_t st);
template <class R>
void ss_launch_active_blocks_scan(const SSDevT<R>& P, const uint32_t* block_flag, uint32_t nblocks, uint32_t cap, uint32_t* list, uint32_t* slot, uint32_t* xyz, uint32_t* state, SSMailSlot mail, hipStream_t st);
template <class R>
void ss_launch_mc_blocks_scan(const SSDevT<R>& P, const uint32_t* block_slot, const ss_real2<R>* blk_minmax, uint32_t nblocks, uint32_t cap, uint32_t* mc_list, uint32_t* mc_slot, uint32_t* mc_xyz, uint32_t* state, SSMailSlot mail, hipStream_t st);
// state2 == nullptr: one packed scan, mail = vertices | triangles << 31 (needs n_mc * SS_MC_MAX_TRI_PER_BLOCK < 2^31); otherwise two scans, mail = vertices, mail2 = triangles
#define SS_MC_MAX_TRI_PER_BLOCK 2560u  // 8^3 cells of at most 5 triangles (and at most 3 * 8^3 = 1536 owned edge vertices)
void ss_launch_mc_offsets_scan(const uint32_t* vcount, const uint32_t* tcount, uint32_t n_mc, uint32_t* vbase, uint32_t* tbase, uint32_t* state, uint32_t* state2, SSMailSlot mail,
                               SSMailSlot mail2, hipStream_t st);
void ss_launch_tile_offsets_scan(const uint32_t* bound, uint32_t n, unsigned long long* off, uint32_t* state, SSMailSlot mail, hipStream_t st);
template <class R>
void ss_launch_posvol_by_index(uint32_t n, const ss_real4<R>* posvol, const uint32_t* perm, ss_real4<R>* posvol_by_index, hipStream_t st);
