// ss_host.h -- host-side state shared by the translation units of libsplashsurf_hip.so (ss_api.hip, ss_post.hip):
// grow-only device/pinned buffers, the context (= the reference's thread pool + workspace.rs) and the result object.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "../../include/splashsurf_hip.h"
#include "ss_device.h"
#include "ss_global.h"
#include "ss_prims.h"

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
#ifdef SS_HIP_EMU  // tests/emu (the kernels on the CPU): exact sizes, so that its AddressSanitizer build sees a kernel that runs past what was asked for
        size_t want = bytes;
#else
        size_t want = bytes + bytes / 8 + 256;
#endif
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ss_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::string err;
    int err_detail = 0;
    // scratch (grow-only, reused across calls = the reference's workspace.rs)
    DevBuf xyz_in, xyz_filt, flags32, offsets, keys_a, keys_b, vals_a, cell_count, cell_start, pos_sorted, temp, aabb_partial, aabb_out, vcount, tcount, counter;
    // per-subdomain particle copies for the density stage
    DevBuf nb_count, nb_tmp;
    DevBuf own_flag;  // per subdomain copy: owned flag, its scan, the list of owned copies
    bool split_mc_offsets = false;  // SS_OPTION_SPLIT_MC_OFFSETS: always two 64-bit scans for the vertex / triangle offsets (tests)
    bool widen_on_device = false;  // SS_OPTION_WIDEN_ON_DEVICE: ss_result_triangles widens the u32 indices on the device and copies u64 (tests; the default for small meshes)
    DevBuf copy_offset, sub_rank, occ_sub, ckeys_a, ckeys_b, cvals_a, cidx, cpos, cell_start2;
    hipEvent_t ev[26];  // (22, 23: around the K1 chain on the second stream; 24 fork, 25 join)
    // ^ 0..9 stage boundaries, 10/11 start of phase 2, 12..15 inside the splat, 16/17 around the arena-path gather, 18/19 k_density_sub, 20/21 k_mc_count
    // the two-pass splat pays off when enough sub-blocks get certified 'inside' (bulk fluid); thin structures do not -- decided per
    // workload from the previous call's statistics, re-probed now and then
    uint64_t early_key = 0;
    bool early_enabled = true;
    int early_skipped = 0;
    int two_pass = -1;           // SS_OPTION_SPLAT_TWO_PASS: -1 automatic, 0 never, 1 always
    bool full_levelset = false;  // SS_OPTION_FULL_LEVELSET: no early exit in the splat (complete level-set values everywhere)
    // exhaustively verified "division by h via reciprocal + 2 FMA" (ss_kernels.hip ss_div_by_h)
    DevBuf fastdiv_scratch;
    float fastdiv_h = 0.0f;
    bool fastdiv_ok = false;
    bool ev_ok = false;
    DevBuf gboxes;  // global strategy: stencil boxes per particle chunk
    DevBuf splat_overflow;  // flags / ranks / list of level-set blocks whose tile is ordered by the workgroup-level gather
    DevBuf splat_trunc;  // per active block: truncated flag, needed-by-MC flag, its scan and the list of blocks to complete
    DevBuf mc_nb;  // per MC block: slots and certified masks of its eight level-set blocks
    DevBuf splat_rowtab;    // per (x, y) row of splat cells a block scans: first / one-past-last cell offset (the same for every block: k_splat_row_table); rowtab_key: everything splat_row_cells reads, for the table in the buffer
    struct RowTabKey { int sn1, sk, kdim1, kdim2, real_bytes; float so, se, srho; } rowtab_key = {0, 0, 0, 0, 0, 0.0f, 0.0f, 0.0f};
    DevBuf splat_tile_idx;  // particle index of every arena entry (tiles the wave-per-block kernel orders itself)
    DevBuf splat_tiles, splat_counts, splat_off, splat_bound;  // tile arena (index-ordered candidates of every block), per-block counts, 64-bit offsets, size bounds
    // counts the host waits for arrive in pinned host memory mapped into the device (SSMailSlot, ss_prims.h): 16 slots of {value, seq}
    unsigned long long* mail_host = nullptr;
    unsigned long long* mail_dev = nullptr;
    unsigned long long mail_seq = 0;
    uint64_t host_waits = 0;  // blocking points of the current call (ss_stats::n_host_waits)
    uint64_t call_serial = 0;  // reconstructions started on this context (stamps what a result keeps in the context's scratch)
    // second stream (experiment, SPLASH_K1_OVERLAP=1): the splat-cell sort (K1, bandwidth-bound) beside the density kernel (bound by the vector L1 / VALU).
    // Measured: S10M-tank 9.62 against 9.67 ms per step, S10M-cube 20.9 / 21.2 -- the K1 chain stretches from 0.41 to 1.33 ms, the density kernel from 0.96
    // to 1.00 ms: the device is busy either way.  Off by default (one stream, K1 first).
    hipStream_t stream2 = nullptr;
    bool overlap_k1 = false;
    DevBuf zeros_k1;  // the zeroed words of the K1 chain (its own memset, on the stream the chain runs on)
    DevBuf zeros;     // zero-initialised words of one phase: scan states, counters (one memset per phase)
    DevBuf sort_work; // work buffer of ss_radix_sort_pairs
    uint32_t cap_active = 0, cap_mc = 0;  // capacities of the block lists (grow-only; a call that needs more repeats the scan that fills them)
    // post-processing: grow-only scratch slots handed out in call order (reset at the start of every ss_post_* call)
    DevBuf post_pool[24];
    int post_pool_next = 0;
};

struct ss_result {
    ss_context* ctx = nullptr;
    bool valid = false;
    bool is_f64 = false;  // Real type of the reconstruction held by this result
    SSDevT<float> P32;
    SSDevT<double> P64;
    SSGlobT<float> Q32;   // parameters of the global strategy (valid when global_strategy)
    SSGlobT<double> Q64;
    ss_grid_f32 grid32, sub32;
    ss_grid_f64 grid64, sub64;
    bool has_inside = false;
    uint64_t n_input = 0, n_particles = 0, n_vertices = 0, n_triangles = 0;
    uint32_t n_active = 0, n_mc = 0;
    const uint32_t* dbg_certified = nullptr;  // per active block: the certified, never evaluated sub-blocks (context scratch of the last call; ss_result_debug_certified)
    uint64_t dbg_serial = 0;                  // ctx->call_serial of the call that set dbg_certified
    bool has_neighbors = false;
    uint64_t n_neighbors = 0;
    DevBuf nb_ptr, nb_idx, nb_idx64;
    HostBuf h_nb_ptr, h_nb_idx;
    bool hnbp = false, hnbi = false;
    bool global_strategy = false;  // reconstruct_surface_global ran: subdomain_grid is None, G is dense over grid.n_points
    int phase = 0;  // 0 nothing, 1 after phase_begin, 2 complete
    bool host_input = false;
    bool density_kernel_timed = false;  // events 18/19 were recorded by this call (they are not when no particle has a copy)
    uint64_t n_occupied_subdomains = 0, n_subdomain_particles = 0;
    ss_stats stats;
    // device results
    DevBuf rho, posvol, posvol_by_index, perm, inside8, G, blk_minmax, block_slot, active_xyz, mc_xyz, mc_slot, masks, vbase, tbase, vertices, vkeys, tri32, tri64;
    // host mirrors
    HostBuf h_vertices, h_tri64, h_tri32, h_rho, h_vkeys, h_inside;
    bool hv = false, ht64 = false, ht32 = false, hrho = false, hkeys = false, hinside = false;
};

#define SS_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            (ctx)->err = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call;            \
            (void)hipGetLastError();                                                                   \
            return SS_ERR_DEVICE;                                                                      \
        }                                                                                              \
    } while (0)

inline ss_status fail(ss_context* ctx, ss_status st, const std::string& msg, int detail = 0) {
    if (ctx) {
        ctx->err = msg;
        ctx->err_detail = detail;
    }
    return st;
}


// hipMemsetAsync of a size that is not a multiple of 16 bytes costs TWO fill launches on ROCm (the aligned body and a tail: config 1 spent 10 fill
// launches on its 6 memsets, profiles/r06_cfg_pmc.md); every zero region of the host flow is therefore reserved and cleared in whole 16-byte units
inline size_t ss_round16(size_t bytes) { return (bytes + 15) & ~(size_t)15; }

inline bool is_device_pointer(const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
