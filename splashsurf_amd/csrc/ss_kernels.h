// ss_kernels.h -- launch wrappers of the gfx950 kernels (ss_kernels.hip), called from ss_api.hip.
#pragma once
#include "ss_device.h"

void ss_launch_aabb(const float* d_xyz, uint32_t n, float* d_partial, float* d_out6, hipStream_t st);
void ss_launch_inside_flags(const float* d_xyz, uint32_t n, const float amin[3], const float amax[3], uint8_t* f8, uint32_t* f32, hipStream_t st);
void ss_launch_compact_xyz(const float* d_xyz, uint32_t n, const uint32_t* f32, const uint32_t* offs, float* out, hipStream_t st);
void ss_launch_cell_keys(const SSDev& P, const float* d_xyz, uint32_t* keys, uint32_t* vals, uint32_t* cell_count, hipStream_t st);
void ss_launch_gather_sorted(uint32_t n, const float* d_xyz, const uint32_t* perm, float4* pos_sorted, hipStream_t st);
void ss_launch_classify_count(const SSDev& P, const float* xyz, uint32_t* member_count, uint32_t* sub_flag, hipStream_t st);
void ss_launch_occupied_list(const uint32_t* flag, const uint32_t* rank, uint32_t n, uint32_t* occ_sub, hipStream_t st);
void ss_launch_emit_copies(const SSDev& P, const float* xyz, const uint32_t* copy_offset, const uint32_t* occ_rank, uint32_t* keys, uint32_t* vals,
                           uint32_t* cell_count, hipStream_t st);
void ss_launch_density_sub(const SSDev& P, uint32_t n_copies, const float4* cpos, const uint32_t* cidx, const uint32_t* ckey,
                           const uint32_t* cell_start, const uint32_t* occ_sub, float* rho, int mode, uint32_t* nb_count,
                           const unsigned long long* nb_ptr, uint32_t* nb_idx, hipStream_t st);
void ss_launch_make_posvol(const SSDev& P, const float4* pos_sorted, const uint32_t* perm, const float* rho, float4* posvol, hipStream_t st);
void ss_launch_mark_blocks(const SSDev& P, const uint32_t* cell_start, uint32_t ncells, uint32_t* block_flag, hipStream_t st);
void ss_launch_mark_mc_blocks(const SSDev& P, const uint32_t* block_slot, const float2* blk_minmax, uint32_t nblocks, uint32_t* mc_flag,
                              hipStream_t st);
void ss_launch_compact_blocks(const uint32_t* flag, const uint32_t* rank, uint32_t nblocks, uint32_t* list, uint32_t* slot, hipStream_t st);
void ss_launch_splat(const SSDev& P, const float4* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_list,
                     uint32_t n_active, float* G, float2* blk_minmax, unsigned long long* cand_counter, bool fast_div, hipStream_t st);
void ss_launch_verify_fast_div(float h, float rh, uint32_t* bad, hipStream_t st);
void ss_launch_mc_count(const SSDev& P, const float* G, const uint32_t* block_slot, const uint32_t* mc_list, uint32_t n_mc,
                        unsigned long long* masks, uint32_t* vcount, uint32_t* tcount, hipStream_t st);
void ss_launch_mc_emit(const SSDev& P, const float* G, const uint32_t* block_slot, const uint32_t* mc_list, const uint32_t* mc_slot,
                       uint32_t n_mc, const unsigned long long* masks, const uint32_t* vbase, const uint32_t* tbase, float* vertices,
                       unsigned long long* vkeys, uint32_t* triangles, hipStream_t st);
void ss_launch_widen(const uint32_t* in, size_t n, unsigned long long* out, hipStream_t st);
void ss_launch_levelset_box(const SSDev& P, const float* G, const uint32_t* block_slot, const int lo[3], const int ext[3], float* out,
                            hipStream_t st);
