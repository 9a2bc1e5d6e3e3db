// ss_api.hip -- C ABI (include/splashsurf_hip.h) and host orchestration of the gfx950 kernels.
//
// Host-side restatement of the reference's grid set-up (lib.rs:476-516, density_map.rs:551-580,
// uniform_grid.rs:175-232, dense_subdomains.rs:89-244) in IEEE f32 (this file is compiled with
// -ffp-contract=off like the kernels).  The radix sort and the prefix sums are the library's own
// (ss_prims.h), every domain kernel is hand-written in ss_kernels.hip.
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <iterator>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/splashsurf_hip.h"
#include "ss_device.h"
#include "ss_kernels.h"
#include "ss_global.h"
#include "ss_host.h"

namespace {

// C structs per Real type
template <class R> struct TypesOf;
template <> struct TypesOf<float> { using params = ss_params_f32; using grid = ss_grid_f32; using shard = ss_shard_f32; };
template <> struct TypesOf<double> { using params = ss_params_f64; using grid = ss_grid_f64; using shard = ss_shard_f64; };
template <class R> SSDevT<R>& dev_params(ss_result* r);
template <> SSDevT<float>& dev_params<float>(ss_result* r) { return r->P32; }
template <> SSDevT<double>& dev_params<double>(ss_result* r) { return r->P64; }
template <class R> typename TypesOf<R>::grid& result_grid(ss_result* r);
template <> ss_grid_f32& result_grid<float>(ss_result* r) { return r->grid32; }
template <> ss_grid_f64& result_grid<double>(ss_result* r) { return r->grid64; }
template <class R> typename TypesOf<R>::grid& result_subgrid(ss_result* r);
template <> ss_grid_f32& result_subgrid<float>(ss_result* r) { return r->sub32; }
template <> ss_grid_f64& result_subgrid<double>(ss_result* r) { return r->sub64; }

// ---- uniform grid, host restatement (uniform_grid.rs:175-232, 647-674) ----
template <class R>
void grid_new(typename TypesOf<R>::grid* g, const R mn[3], const int64_t nc[3], R cs) {
    for (int d = 0; d < 3; ++d) {
        g->aabb_min[d] = mn[d];
        g->n_cells[d] = nc[d];
        g->n_points[d] = nc[d] + 1;
        g->aabb_max[d] = mn[d] + cs * (R)(double)nc[d];
    }
    g->cell_size = cs;
}

template <class R>
int grid_from_aabb(typename TypesOf<R>::grid* g, const R amin[3], const R amax[3], R cs) {
    if (!(cs > R(0.0))) return SS_GRID_INVALID_CELL_SIZE;
    if (amin[0] == amax[0] && amin[1] == amax[1] && amin[2] == amax[2]) return SS_GRID_DEGENERATE_AABB;
    if (!(amin[0] <= amax[0] && amin[1] <= amax[1] && amin[2] <= amax[2])) return SS_GRID_INCONSISTENT_AABB;
    R aligned[3];
    int64_t nc[3];
    for (int d = 0; d < 3; ++d) {
        aligned[d] = ss_floor(amin[d] / cs) * cs;
        R n_real = (amax[d] - aligned[d]) / cs;
        double c = (double)ss_ceil(n_real);
        if (!(c < 2147483000.0)) return SS_GRID_INDEX_TYPE_TOO_SMALL;  // this build indexes points with i32 per dimension
        int64_t n = (int64_t)c;
        nc[d] = n < 1 ? 1 : n;
    }
    grid_new(g, aligned, nc, cs);
    return 0;
}

// lib.rs:476-516 given the particle AABB (already computed on the device or supplied by the user)
template <class R>
int grid_for_reconstruction(const typename TypesOf<R>::params* prm, bool have_particles, const R pmin[3], const R pmax[3], typename TypesOf<R>::grid* out) {
    R amin[3], amax[3];
    if (prm->has_particle_aabb) {
        for (int d = 0; d < 3; ++d) {
            amin[d] = prm->aabb_min[d];
            amax[d] = prm->aabb_max[d];
        }
    } else {
        for (int d = 0; d < 3; ++d) {
            amin[d] = have_particles ? pmin[d] : R(0.0);  // aabb.rs:28-31: empty -> zeros
            amax[d] = have_particles ? pmax[d] : R(0.0);
            amin[d] -= prm->particle_radius;  // lib.rs:496
            amax[d] += prm->particle_radius;
        }
    }
    const R half_cells = ss_ceil(prm->compact_support_radius / prm->cube_size);  // density_map.rs:563
    const R eps_sqrt = ss_sqrt(std::numeric_limits<R>::epsilon());
    const R kernel_margin = prm->cube_size * half_cells * (R(1.0) + eps_sqrt);  // density_map.rs:572-573
    for (int d = 0; d < 3; ++d) {
        amin[d] -= kernel_margin;  // lib.rs:513
        amax[d] += kernel_margin;
    }
    return grid_from_aabb(out, amin, amax, prm->cube_size);
}

// dense_subdomains.rs:89-244
template <class R>
void initialize_subdomain_parameters(const typename TypesOf<R>::params* prm, const typename TypesOf<R>::grid* initial, typename TypesOf<R>::grid* global_grid, typename TypesOf<R>::grid* sub_grid,
                                     R* mass, R* margin) {
    const int64_t n = (int64_t)prm->subdomain_num_cubes_per_dim;
    const R d = prm->particle_radius + prm->particle_radius;  // kernel.rs:28-30
    *mass = (d * d * d) * prm->rest_density;
    *margin = ss_ceil(prm->compact_support_radius / prm->cube_size) * prm->cube_size * R(1.01);
    int64_t nsub[3], ncell[3];
    for (int k = 0; k < 3; ++k) {
        int64_t c = initial->n_cells[k];
        int64_t rem = c % n;
        nsub[k] = c / n + (rem < 1 ? rem : 1);  // int_ceil_div, :2129-2131
        ncell[k] = nsub[k] * n;
    }
    grid_new(global_grid, initial->aabb_min, ncell, prm->cube_size);
    const R sub_size = prm->cube_size * (R)(double)n;
    grid_new(sub_grid, global_grid->aabb_min, nsub, sub_size);
}


// plain exclusive prefix sum of an array (the secondary paths: global strategy, particle AABB filter, neighbour lists); the hot path
// uses the fused forms of ss_kernels.h
template <class T>
struct ArrayIn {
    const T* p;
    __device__ T operator()(uint32_t i) const { return p[i]; }
};
template <class T>
struct ArrayExclOut {
    T* p;
    __device__ void operator()(uint32_t i, T, T excl) const { p[i] = excl; }
};
template <class T>
ss_status exclusive_scan_u32(ss_context* ctx, const T* in, T* out, size_t n) {
    if (n > SS_SCAN_MAX_N) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 - 8193 entries in one prefix sum");
    const size_t words = ss_scan_state_words(n);
    SS_HIP(ctx, ctx->temp.reserve(ss_round16(words * 4)));
    SS_HIP(ctx, hipMemsetAsync(ctx->temp.p, 0, ss_round16(words * 4), ctx->stream));
    ss_chained_scan<T, SSOpPlus>(ArrayIn<T>{in}, ArrayExclOut<T>{out}, (uint32_t)n, ctx->temp.as<uint32_t>(), (T*)nullptr, SSMailSlot{}, ctx->stream);
    return SS_OK;
}

// ---- counts the host waits for (SSMailSlot, ss_prims.h) ----
#define SS_MAIL_SLOTS 32  // 0-10 counts of the phases, 11 + 12..15 the particle AABB, 16 the triangle total of the split offsets scan
ss_status ensure_mail(ss_context* ctx) {
    if (ctx->mail_host) return SS_OK;
    void* h = nullptr;
    SS_HIP(ctx, hipHostMalloc(&h, SS_MAIL_SLOTS * 2 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(h, 0, SS_MAIL_SLOTS * 2 * sizeof(unsigned long long));
    void* d = nullptr;
    SS_HIP(ctx, hipHostGetDevicePointer(&d, h, 0));
    ctx->mail_host = reinterpret_cast<unsigned long long*>(h);
    ctx->mail_dev = reinterpret_cast<unsigned long long*>(d);
    return SS_OK;
}
SSMailSlot mail_slot(ss_context* ctx, int k) { return SSMailSlot{ctx->mail_dev + 2 * k, ++ctx->mail_seq}; }
// Waits until the kernel holding `m` has posted: the host polls the pinned word instead of synchronising the stream (the stream goes on with
// whatever was enqueued behind that kernel).  A drained stream without the value, a stream error or 120 s end the wait with an error.
// `grouped`: the slot is posted together with the one waited for just before (one blocking point, counted once)
ss_status mail_wait(ss_context* ctx, const SSMailSlot& m, unsigned long long* value, bool grouped = false) {
    if (!grouped) ++ctx->host_waits;
    const int k = (int)((m.p - ctx->mail_dev) / 2);
    volatile unsigned long long* h = ctx->mail_host + 2 * k;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long it = 0;; ++it) {
        if (__atomic_load_n(&h[1], __ATOMIC_ACQUIRE) == m.seq) {
            *value = h[0];
            return SS_OK;
        }
        if ((it & 0xFFFu) == 0xFFFu) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {
                if (__atomic_load_n(&h[1], __ATOMIC_ACQUIRE) == m.seq) {
                    *value = h[0];
                    return SS_OK;
                }
                return fail(ctx, SS_ERR_DEVICE, "a count the host waits for never arrived (stream drained)");
            }
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                return fail(ctx, SS_ERR_DEVICE, std::string("HIP error while waiting for a count: ") + hipGetErrorString(q));
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
                return fail(ctx, SS_ERR_DEVICE, "timed out waiting for a count from the device");
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}
// n zeroed 32-bit words from the context's zero region (one memset per phase: reserve_zeros first, then take)
struct ZeroTaker {
    uint32_t* base = nullptr;
    size_t used = 0, cap = 0;
    uint32_t* take(size_t words) {
        words = (words + 3) & ~(size_t)3;
        uint32_t* p = base + used;
        used += words;
        return used <= cap ? p : nullptr;
    }
};
ss_status reserve_zeros(ss_context* ctx, size_t words, ZeroTaker* z) {
    words = (words + 64 + 3) & ~(size_t)3;  // (whole 16-byte units: ss_round16)
    SS_HIP(ctx, ctx->zeros.reserve(words * 4));
    SS_HIP(ctx, hipMemsetAsync(ctx->zeros.p, 0, words * 4, ctx->stream));
    z->base = ctx->zeros.as<uint32_t>();
    z->used = 0;
    z->cap = words;
    return SS_OK;
}
// stable sort of the (key, position) pairs by the low `bits` bits; returns the buffers that hold the result
// `zero_work`: zeroed words for the sort (ss_radix_sort_work_words) from the caller's zero region; null: the sort's own buffer, zeroed here
ss_status sort_pairs(ss_context* ctx, uint32_t* keys[2], uint32_t* vals[2], uint32_t n, unsigned bits, bool iota, uint32_t* zero_work, int* result, hipStream_t st = nullptr) {
    if (n >= (1u << 30)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^30 - 1 entries to sort in one call are not supported by this build");
    if (zero_work) {
        *result = ss_radix_sort_pairs(keys, vals, n, bits, iota, zero_work, true, st ? st : ctx->stream);
        return SS_OK;
    }
    SS_HIP(ctx, ctx->sort_work.reserve(ss_radix_sort_work_words(n, bits) * 4));
    *result = ss_radix_sort_pairs(keys, vals, n, bits, iota, ctx->sort_work.as<uint32_t>(), false, ctx->stream);
    return SS_OK;
}

template <class PRM>
ss_status validate_params(ss_context* ctx, const PRM* prm, uint64_t n) {
    if (!prm) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "parameters pointer is null");
    if (n >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^31-1 particles per call are not supported by this build");
    // the reference panics for these (density_map.rs:555-559); report instead of aborting the host
    if (!(prm->cube_size > 0.0f)) return fail(ctx, SS_ERR_UNKNOWN, "cube size must be positive (reference: panic in compute_kernel_evaluation_radius)");
    if (!(prm->compact_support_radius >= 0.0f)) return fail(ctx, SS_ERR_UNKNOWN, "compact support radius must be non-negative");
    if (!(prm->compact_support_radius > 0.0f)) return fail(ctx, SS_ERR_UNKNOWN, "search radius for neighborhood search has to be positive");
    if (prm->subdomain_num_cubes_per_dim < 1 || prm->subdomain_num_cubes_per_dim > (1u << 20))
        return fail(ctx, SS_ERR_INVALID_ARGUMENT, "subdomain_num_cubes_per_dim out of range");
    return SS_OK;
}

void reset_host_flags(ss_result* r) { r->hv = r->ht64 = r->ht32 = r->hrho = r->hkeys = r->hinside = r->hnbp = r->hnbi = false; }

template <class R>
ss_status make_device_params(ss_context* ctx, const typename TypesOf<R>::params* prm, const typename TypesOf<R>::grid& g, const typename TypesOf<R>::grid& sg, R mass, R margin,
                             uint32_t n, const typename TypesOf<R>::shard* shard, SSDevT<R>* out) {
    SSDevT<R> P;
    memset(&P, 0, sizeof(P));
    const R h = prm->compact_support_radius;
    for (int d = 0; d < 3; ++d) {
        P.gmin[d] = g.aabb_min[d];
        P.np[d] = (int)g.n_points[d];
        P.nc[d] = (int)g.n_cells[d];
        P.ns[d] = (int)sg.n_cells[d];
        P.nb[d] = (P.np[d] + SS_BLOCK - 1) / SS_BLOCK;
    }
    P.cs = g.cell_size;
    P.n_sub_cubes = (int)prm->subdomain_num_cubes_per_dim;
    P.sub_size = sg.cell_size;
    P.sub_radius = (int)ss_ceil(margin / sg.cell_size);  // dense_subdomains.rs:1827-1832
    if (P.sub_radius < 1 || P.sub_radius > 64) return fail(ctx, SS_ERR_UNSUPPORTED, "ghost margin spans more than 64 subdomains");
    for (int d = 0; d < 3; ++d) P.sc[d] = (int)ceil(((double)sg.cell_size + 3.0 * (double)margin) / (double)h) + 3;
    P.h = h;
    P.inv_h = 1.0 / (double)h;
    P.h2 = h * h;
    P.H2 = (h * h) * R(1.01);
    P.sigma = R(8.0) / (h * h * h);
    P.w0 = ss_kernel_evaluate(R(0.0), h, P.sigma);
    P.mass = mass;
    P.threshold = prm->iso_surface_threshold;
    P.margin = margin;
    // Parameters::enable_simd (lib.rs:179-181): the SIMD loop of the reference exists for <i64, f32> only (dense_subdomains.rs:1413-1415)
    const int simd = (sizeof(R) == 4) ? (prm->enable_simd == 2 ? 2 : (prm->enable_simd != 0 ? 1 : 0)) : 0;
    P.arith = simd == 2 ? SS_ARITH_SIMD_HW : (simd == 1 ? SS_ARITH_SIMD : SS_ARITH_GENERIC);  // refined by choose_arith() once the device checks ran
    // support of a particle on the grid: d^2 < 1.01 h^2 in the scalar loop (:1224-1226, :831), d^2 < h^2 in the SIMD loop (:1037-1038, :1083)
    const R support_factor = simd ? R(1.0) : R(1.01);
    P.reach = ss_sqrt(support_factor) * h * R(1.0001);
    P.R2 = ((h * h) * support_factor) * R(1.0001);
#ifndef SS_TUNE_RNEAR
#define SS_TUNE_RNEAR 0.64
#endif
    // near radius of the classification pass, measured on S10M-tank with the polynomial bound u^3 (c0 + c1 u^2) (splat kernel ms /
    // certified sub-blocks): 0.50 h 9.87 / 74 %, 0.52 h 8.69 / 80 %, 0.55 h 7.73 / 85 %, 0.58 h 7.40 / 87 %, 0.60 h 7.50 / 87 % with
    // per-sub-block lists of f32 records; with the pooled f16 records (splat_bound_record): 0.56 h 6.35 / 85.6 %, 0.58 h 6.33 / 86.6 %,
    // 0.60 h 6.26 / 87.2 %, 0.62 h 6.59 / 87.5 %, 0.65 h 7.23 / 87.6 % (the lists outgrow the pool) -- an uncertified sub-block costs
    // five times its classification, so the optimum sits where the curve flattens.
    // (Round 2's bound v^2 min(2 v, 1) with its v_sqrt_f32: 8.17 ms / 86 % at 0.60 h.)
    // Round 6, the certificate on the matrix pipe with the bound C4 u^4 (90 % of the kernel's mass instead of 96 %; an entry of a list costs a
    // third of what it did, whole tiles of 32 rows cost the same up to the next eight rows): 0.60 h 4.64-4.68 ms / 86.9 %, 0.62 h 4.32-4.38 / 87.8 %,
    // 0.64 h 4.27-4.28 / 88.1 %, 0.66 h 4.30-4.31 / 88.2 %, 0.68 h 4.32-4.34 / 88.3 % (profiles/r06_ab_mfma_certificate.jsonl).
    P.R2near = (R(SS_TUNE_RNEAR) * h) * (R(SS_TUNE_RNEAR) * h);
    P.thr_inside = prm->iso_surface_threshold * R(1.0001);
    R amax = R(0.0);
    for (int d = 0; d < 3; ++d) amax = ss_max(amax, ss_max(std::fabs(g.aabb_min[d]), std::fabs(g.aabb_max[d])));
    P.coord_slack = R(16.0) * std::numeric_limits<R>::epsilon() * amax + 1e-30f;
    {   // splat_cert_record (ss_kernels.hip): slack of the f16 operands of the certificate's tiles.  xm: largest |coordinate| of a block's points
        // relative to the block's centre, in units of h.
        const double xm = 3.5 * (double)prm->cube_size / (double)h * (1.0 + 1.0e-5) + 1.0e-6;
        const double r = std::ldexp(1.0, -10) * (1.0 + std::ldexp(1.0, -10));
        P.cert_e1 = (R)(r * 2.0 * xm * (1.0 + 1.0e-6));
        P.cert_e0 = (R)((r * 3.0 * xm * xm + 3.0e-5) * (1.0 + 1.0e-6));
        const double xs = 1.5 * (double)prm->cube_size / (double)h * (1.0 + 1.0e-5) + 1.0e-6;  // relative to a sub-block's centre
        P.cert_e1s = (R)(r * 2.0 * xs * (1.0 + 1.0e-6));
        P.cert_e0s = (R)((r * 3.0 * xs * xs + 3.0e-5) * (1.0 + 1.0e-6));
    }
    {   // CubicSplineKernelAvxF32::new (kernel.rs:327-337), in f32 like the reference
        const float hf = (float)h;
        const float pi_f = 3.14159265358979323846f;
        const float sig = 8.0f / (pi_f * (hf * hf * hf));
        P.avx_inv_h = (R)(1.0f / hf);
        P.avx_sigma = (R)sig;
        P.avx_sigma2 = (R)(2.0f * sig);
        P.avx_sigma6 = (R)(6.0f * sig);
        P.avx_sigma12 = (R)(12.0f * sig);
        P.cert_vscale = (R)((double)0.76293 * (double)sig * (1.0 - 2.0e-5));  // C4 sigma: the bound C4 u^4 <= W / sigma (SS_CERT_C4), rounded down
    }
    double ncells = 1.0, nblocks = 1.0;
    {   // splat cells (ss_device.h): edge e = 8 / sk grid cells, aligned with the block lattice.  In units of cs relative to a block's first
        // point the block's points span [0, 7]; dilated by rho (reach + rounding slack, padded) the box has width W = 7 + 2 rho and is
        // covered by n1 = floor(W / e) + 1 cells whose first one starts at o = -rho - (n1 e - W) / 2 (the slack split evenly, so that a
        // particle whose cell is off by rounding still lies inside).  sk: the finest subdivision that keeps the n1^2 rows of a block within
        // one wave (64) and the cell edge at about h or above.  Measured on S10M-tank (cs = h / 8): sk = 1 (cells of edge h, 9 rows, ~216
        // candidates per block for 142 within reach; 290 with the former cells of edge h aligned at the origin) and sk = 2 (edge h / 2, 36
        // rows) give the same splat kernel, 5.97 against 6.23 ms, but sk = 2 has eight times the cells: sort, cell table and k_mark_blocks
        // cost 0.6 ms more.  At cs = h / 2 (R = 2) sk = 1 would make cells of edge 4 h: gather 0.64 against 0.43 ms with sk = 5.
        const double cs = (double)g.cell_size;
        const double reach_c = ((double)P.reach + (double)P.coord_slack) / cs;
#ifndef SS_TUNE_SK_MAX
#define SS_TUNE_SK_MAX 8
#endif
        int k_want = (int)floor(8.0 * cs / (double)h + 0.5);
        k_want = std::max(1, std::min(k_want, SS_TUNE_SK_MAX));
        int best_k = 1, best_n1 = 0;
        double best_o = 0.0, best_rho = 0.0;
        for (int k = 1; k <= k_want; ++k) {
            const double e = 8.0 / (double)k;
            const double rho = reach_c * (1.0 + 1.0e-6) + 2.0e-3 * e + 1.0e-6;
            const double W = 7.0 + 2.0 * rho;
            int n1 = (int)floor(W / e) + 1;
            if ((double)n1 * e - W < 1.0e-6 * e) ++n1;
            if (k > 1 && n1 > 8) break;
            best_k = k;
            best_n1 = n1;
            best_rho = rho;
            best_o = -rho - 0.5 * ((double)n1 * e - W);
        }
        if (best_n1 > 4000) return fail(ctx, SS_ERR_UNSUPPORTED, "compact support radius spans too many grid cells for this build");
        P.sk = best_k;
        P.sn1 = best_n1;
        P.so = (float)best_o;
        P.se = (float)(8.0 / (double)best_k);
        P.srho = (float)best_rho;
        P.sinv = (double)best_k / (8.0 * cs);
        for (int d = 0; d < 3; ++d) P.sorg[d] = (double)g.aabb_min[d] + best_o * cs;
    }
    P.n = n;
    // shard region
    bool full = true;
    for (int d = 0; d < 3; ++d) {
        int64_t lo = 0, hi = P.ns[d];
        if (shard) {
            lo = shard->sub_lo[d];
            hi = shard->sub_hi[d];
            if (lo < 0 || hi > P.ns[d] || lo > hi) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "shard subdomain range outside the subdomain grid");
        }
        if (lo != 0 || hi != P.ns[d]) full = false;
        P.sub_lo[d] = (int)lo;
        P.sub_hi[d] = (int)hi;
    }
    if (!full && (P.n_sub_cubes % SS_BLOCK) != 0)
        return fail(ctx, SS_ERR_UNSUPPORTED, "sharded reconstruction needs subdomain_num_cubes_per_dim to be a multiple of 8");
    for (int d = 0; d < 3; ++d) {
        if (P.sub_lo[d] >= P.sub_hi[d]) {  // empty shard: nothing to reconstruct
            P.pt_lo[d] = 0;
            P.pt_hi[d] = -1;
            P.blk_lo[d] = 0;
            P.blk_hi[d] = -1;
            continue;
        }
        P.pt_lo[d] = P.sub_lo[d] * P.n_sub_cubes;
        int hi = P.sub_hi[d] * P.n_sub_cubes;
        if (hi > P.np[d] - 1) hi = P.np[d] - 1;
        P.pt_hi[d] = hi;
        P.blk_lo[d] = P.pt_lo[d] / SS_BLOCK;
        P.blk_hi[d] = P.pt_hi[d] / SS_BLOCK;
    }
    // The dense tables cover what this process touches: the blocks [blk_lo, blk_hi] plus the layer above that marching cubes looks into,
    // and the search cells within reach of those blocks' points (the particles a shard holds lie within the ghost margin of its brick:
    // inside that range; ss_particle_cell clamps).  For a single-process job that is the whole grid.
    for (int d = 0; d < 3; ++d) {
        const bool empty = P.blk_hi[d] < P.blk_lo[d];
        P.bt_org[d] = empty ? 0 : P.blk_lo[d];
        P.bt_dim[d] = empty ? 1 : std::min(P.blk_hi[d] + 1, P.nb[d] - 1) - P.blk_lo[d] + 1;
        // the splat cells covering the blocks of the table window: block b <-> cells [sk b, sk b + sn1)
        const double kd = (double)P.sk * (double)(P.bt_dim[d] - 1) + (double)P.sn1;
        if (!(kd < 2.0e9)) return fail(ctx, SS_ERR_UNSUPPORTED, "splat cell index out of i32 range");
        P.kmin[d] = P.sk * P.bt_org[d];
        P.kdim[d] = (int)kd;
        ncells *= (double)P.kdim[d];
        nblocks *= (double)P.bt_dim[d];
    }
    if (ncells > 4.0e9 || nblocks > 4.0e9)
        return fail(ctx, SS_ERR_UNSUPPORTED, "domain too large for the dense cell/block tables of this build (> 4e9 search cells or level-set blocks)");
    *out = P;
    return SS_OK;
}

ss_status ensure_events(ss_context* ctx) {
    if (ctx->ev_ok) return SS_OK;
    for (int i = 0; i < 26; ++i) SS_HIP(ctx, hipEventCreate(&ctx->ev[i]));
    ctx->ev_ok = true;
    return SS_OK;
}

float ev_ms(ss_context* ctx, int a, int b) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) {
        (void)hipGetLastError();
        return 0.0f;
    }
    return ms;
}

// Uploads (if needed) and filters the particles; returns device pointer to the particles used
template <class R>
ss_status stage_particles(ss_context* ctx, const R* xyz, uint64_t n_in, const typename TypesOf<R>::params* prm, ss_result* res, const R** d_used,
                          uint32_t* n_used) {
    hipStream_t st = ctx->stream;
    const R* d_xyz = nullptr;
    if (n_in == 0) {
        *d_used = nullptr;
        *n_used = 0;
        if (res) {
            res->has_inside = prm->has_particle_aabb != 0;
        }
        return SS_OK;
    }
    if (!xyz) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "particle pointer is null");
    if (is_device_pointer(xyz)) {
        d_xyz = xyz;
    } else {
        SS_HIP(ctx, ctx->xyz_in.reserve(n_in * 3 * sizeof(R)));
        SS_HIP(ctx, hipMemcpyAsync(ctx->xyz_in.p, xyz, n_in * 3 * sizeof(R), hipMemcpyHostToDevice, st));
        d_xyz = ctx->xyz_in.as<R>();
    }
    if (!prm->has_particle_aabb) {
        *d_used = d_xyz;
        *n_used = (uint32_t)n_in;
        if (res) res->has_inside = false;
        return SS_OK;
    }
    // lib.rs:369-406
    DevBuf local_inside;
    DevBuf* inside = res ? &res->inside8 : &local_inside;
    SS_HIP(ctx, inside->reserve(n_in));
    SS_HIP(ctx, ctx->flags32.reserve(ss_round16((n_in + 1) * 4)));
    SS_HIP(ctx, ctx->offsets.reserve((n_in + 1) * 4));
    SS_HIP(ctx, hipMemsetAsync(ctx->flags32.p, 0, ss_round16((n_in + 1) * 4), st));
    ss_launch_inside_flags(d_xyz, (uint32_t)n_in, prm->aabb_min, prm->aabb_max, inside->as<uint8_t>(), ctx->flags32.as<uint32_t>(), st);
    ss_status s = exclusive_scan_u32<uint32_t>(ctx, ctx->flags32.as<uint32_t>(), ctx->offsets.as<uint32_t>(), n_in + 1);
    if (s != SS_OK) return s;
    uint32_t cnt = 0;
    SS_HIP(ctx, hipMemcpyAsync(&cnt, ctx->offsets.as<uint32_t>() + n_in, 4, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    ++ctx->host_waits;
    SS_HIP(ctx, ctx->xyz_filt.reserve((size_t)cnt * 3 * sizeof(R) + 16));
    ss_launch_compact_xyz(d_xyz, (uint32_t)n_in, ctx->flags32.as<uint32_t>(), ctx->offsets.as<uint32_t>(), ctx->xyz_filt.as<R>(), st);
    *d_used = ctx->xyz_filt.as<R>();
    *n_used = cnt;
    if (res) res->has_inside = true;
    if (!res) {
        SS_HIP(ctx, hipStreamSynchronize(st));
        local_inside.release();
    }
    return SS_OK;
}

template <class R>
ss_status compute_particle_aabb(ss_context* ctx, const R* d_xyz, uint32_t n, R pmin[3], R pmax[3]) {
    hipStream_t st = ctx->stream;
    SS_HIP(ctx, ctx->aabb_partial.reserve(SS_AABB_PARTIAL_WORDS * sizeof(R)));
    ss_status s = ensure_mail(ctx);
    if (s != SS_OK) return s;
    // the six values land in pinned host memory (mail slots 12..15), announced through slot 11
    const SSMailSlot m = mail_slot(ctx, 11);
    ss_launch_aabb(d_xyz, n, ctx->aabb_partial.as<R>(), reinterpret_cast<R*>(ctx->mail_dev + 2 * 12), m, st);
    unsigned long long not_finite = 0;
    s = mail_wait(ctx, m, &not_finite);
    if (s != SS_OK) return s;
    // (the reference's AABB skips a NaN like ours and then files the particle under cell 0 -- `NaN as i64` -- with undefined consequences; here a
    // non-finite coordinate has no cell at all, so the input is refused; with an explicit particle AABB such particles are filtered out like any other outside it)
    if (not_finite) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "particle coordinates must be finite (the input holds a NaN or an infinity)");
    const volatile R* h6 = reinterpret_cast<const volatile R*>(ctx->mail_host + 2 * 12);
    for (int d = 0; d < 3; ++d) {
        pmin[d] = h6[d];
        pmax[d] = h6[3 + d];
    }
    return SS_OK;
}

// ---- strategy choice (lib.rs:419-462) ----
template <class R>
bool use_global_strategy(const typename TypesOf<R>::params* prm, const typename TypesOf<R>::grid& initial) {
    if (prm->decomposition == 0) return true;  // SpatialDecomposition::None
    if (!prm->auto_disable) return false;
    int64_t max_cubes = std::max(initial.n_cells[0], std::max(initial.n_cells[1], initial.n_cells[2]));
    const uint32_t with_margin = (uint32_t)(1.2 * (double)prm->subdomain_num_cubes_per_dim);
    const uint32_t mc = max_cubes > (int64_t)UINT32_MAX ? UINT32_MAX : (uint32_t)max_cubes;
    return !(mc > with_margin);
}

// marching cubes of the global strategy on the dense level-set array res->G (narrow_band_extraction.rs, triangulation.rs);
// records events 7..9; also serves ss_marching_cubes_* (triangulate_density_map on DensityMap::Dense, marching_cubes.rs:100-127)
template <class R>
ss_status global_marching_cubes(ss_context* ctx, const SSGlobT<R>& Q, ss_result* res, uint64_t* nv_out, uint64_t* nt_out) {
    hipStream_t st = ctx->stream;
    ss_status s = SS_OK;
    const size_t npts = (size_t)Q.np[0] * Q.np[1] * Q.np[2], ncell = (size_t)Q.nc[0] * Q.nc[1] * Q.nc[2];
    SS_HIP(ctx, ctx->counter.reserve(64));
    uint32_t* d_err = ctx->counter.as<uint32_t>();
    SS_HIP(ctx, res->masks.reserve(npts + 16));
    SS_HIP(ctx, ctx->vcount.reserve((npts + 1) * 4));
    SS_HIP(ctx, ctx->tcount.reserve((ncell + 1) * 4));
    SS_HIP(ctx, res->vbase.reserve((npts + 1) * 4));
    SS_HIP(ctx, res->tbase.reserve((ncell + 1) * 4));
    SS_HIP(ctx, hipMemsetAsync(ctx->vcount.as<uint32_t>() + npts, 0, 4, st));
    SS_HIP(ctx, hipMemsetAsync(ctx->tcount.as<uint32_t>() + ncell, 0, 4, st));
    ssg_launch_edge_masks<R>(Q, res->G.as<R>(), res->masks.as<uint8_t>(), ctx->vcount.as<uint32_t>(), st);
    ssg_launch_cell_count<R>(Q, res->G.as<R>(), res->masks.as<uint8_t>(), ctx->tcount.as<uint32_t>(), d_err, st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[7], st));
    s = exclusive_scan_u32<uint32_t>(ctx, ctx->vcount.as<uint32_t>(), res->vbase.as<uint32_t>(), npts + 1);
    if (s != SS_OK) return s;
    s = exclusive_scan_u32<uint32_t>(ctx, ctx->tcount.as<uint32_t>(), res->tbase.as<uint32_t>(), ncell + 1);
    if (s != SS_OK) return s;
    uint32_t totals[2] = {0, 0}, herr = 0;
    SS_HIP(ctx, hipMemcpyAsync(&totals[0], res->vbase.as<uint32_t>() + npts, 4, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(&totals[1], res->tbase.as<uint32_t>() + ncell, 4, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    if (herr & 2u) return fail(ctx, SS_ERR_MARCHING_CUBES, "missing iso surface vertex at an edge (TriangulationError, triangulation.rs:62-95)");
    const uint64_t nv = totals[0], nt = totals[1];
    *nv_out = nv;
    *nt_out = nt;
    if (nt * 3 >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32/3 triangles in one call are not supported by this build");
    SS_HIP(ctx, res->vertices.reserve(nv * 3 * sizeof(R) + 16));
    SS_HIP(ctx, res->vkeys.reserve(nv * 8 + 16));
    SS_HIP(ctx, res->tri32.reserve(nt * 12 + 16));
    SS_HIP(ctx, hipEventRecord(ctx->ev[8], st));
    ssg_launch_emit_vertices<R>(Q, res->G.as<R>(), res->masks.as<uint8_t>(), res->vbase.as<uint32_t>(), res->vertices.as<R>(), res->vkeys.as<unsigned long long>(), st);
    ssg_launch_emit_triangles<R>(Q, res->G.as<R>(), res->masks.as<uint8_t>(), res->vbase.as<uint32_t>(), ctx->tcount.as<uint32_t>(), res->tbase.as<uint32_t>(),
                                 res->tri32.as<uint32_t>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[9], st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ctx, SS_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(e));
    }
    return SS_OK;
}

// neighbourhood search on the grid Q.smin/Q.snc (cell size Q.h) + densities; fills res->rho, res->nb_ptr, res->nb_idx
// (neighborhood_search.rs:148-230, density_map.rs:113-186); records event 3 after the cell map is built
template <class R>
ss_status global_search_and_densities(ss_context* ctx, const SSGlobT<R>& Q, const R* d_xyz, ss_result* res) {
    hipStream_t st = ctx->stream;
    ss_status s = SS_OK;
    const uint32_t n = Q.n;
    const size_t nscell = (size_t)Q.snc[0] * Q.snc[1] * Q.snc[2];
    SS_HIP(ctx, ctx->counter.reserve(64));
    SS_HIP(ctx, hipMemsetAsync(ctx->counter.p, 0, 64, st));
    uint32_t* d_err = ctx->counter.as<uint32_t>();
    SS_HIP(ctx, res->rho.reserve((size_t)n * sizeof(R) + 16));
    SS_HIP(ctx, res->perm.reserve((size_t)n * 4 + 16));
    SS_HIP(ctx, ctx->cell_count.reserve(ss_round16((nscell + 1) * 4)));
    SS_HIP(ctx, ctx->cell_start.reserve((nscell + 1) * 4));
    SS_HIP(ctx, hipMemsetAsync(ctx->cell_count.p, 0, ss_round16((nscell + 1) * 4), st));
    SS_HIP(ctx, res->nb_ptr.reserve(((size_t)n + 1) * 8));
    res->has_neighbors = true;  // the global strategy always returns the neighbour lists (reconstruction.rs:107-108)
    res->n_neighbors = 0;
    if (n > 0) {
        SS_HIP(ctx, ctx->keys_a.reserve((size_t)n * 4));
        SS_HIP(ctx, ctx->keys_b.reserve((size_t)n * 4));
        SS_HIP(ctx, ctx->vals_a.reserve((size_t)n * 4));
        ssg_launch_cell_keys<R>(Q, d_xyz, ctx->keys_a.as<uint32_t>(), ctx->vals_a.as<uint32_t>(), ctx->cell_count.as<uint32_t>(), d_err, st);
        s = exclusive_scan_u32<uint32_t>(ctx, ctx->cell_count.as<uint32_t>(), ctx->cell_start.as<uint32_t>(), nscell + 1);
        if (s != SS_OK) return s;
        unsigned bits = 1;
        while (bits < 32 && ((size_t)1 << bits) < nscell) ++bits;
        {   // stable sort of (cell, particle) by cell; the buffers are assigned so that the sorted particle indices end in res->perm
            const bool odd = (((bits + 7u) / 8u) & 1u) != 0u;
            SS_HIP(ctx, ctx->vals_a.reserve((size_t)n * 4 + 16));
            uint32_t* keys[2] = {ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>()};
            // (the values are the particle indices 0 .. n-1: the sort supplies them itself)
            uint32_t* vals[2] = {odd ? ctx->vals_a.as<uint32_t>() : res->perm.as<uint32_t>(), odd ? res->perm.as<uint32_t>() : ctx->vals_a.as<uint32_t>()};
            int r = 0;
            s = sort_pairs(ctx, keys, vals, n, bits, true, nullptr, &r);
            if (s != SS_OK) return s;
            if (vals[r] != res->perm.as<uint32_t>()) return fail(ctx, SS_ERR_UNKNOWN, "internal error: sort result in an unexpected buffer");
        }
        uint32_t herr = 0;
        SS_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        if (herr & 1u) return fail(ctx, SS_ERR_UNKNOWN, "particle outside the neighborhood-search grid (reference: panic in get_cell().unwrap())");
        SS_HIP(ctx, hipEventRecord(ctx->ev[3], st));
        SS_HIP(ctx, ctx->nb_count.reserve(ss_round16(((size_t)n + 1) * 8)));
        SS_HIP(ctx, ctx->nb_tmp.reserve(((size_t)n + 1) * 8));
        SS_HIP(ctx, hipMemsetAsync(ctx->nb_count.p, 0, ss_round16(((size_t)n + 1) * 8), st));
        ssg_launch_density<R>(Q, d_xyz, ctx->cell_start.as<uint32_t>(), res->perm.as<uint32_t>(), res->rho.as<R>(), 0, ctx->nb_count.as<uint32_t>(), nullptr,
                              nullptr, st);
        ss_launch_widen(ctx->nb_count.as<uint32_t>(), (size_t)n + 1, ctx->nb_tmp.as<unsigned long long>(), st);
        s = exclusive_scan_u32<unsigned long long>(ctx, ctx->nb_tmp.as<unsigned long long>(), res->nb_ptr.as<unsigned long long>(), (size_t)n + 1);
        if (s != SS_OK) return s;
        unsigned long long total_nb = 0;
        SS_HIP(ctx, hipMemcpyAsync(&total_nb, res->nb_ptr.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost, st));
        SS_HIP(ctx, hipStreamSynchronize(st));
        if (total_nb >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 neighbour entries");
        res->n_neighbors = total_nb;
        SS_HIP(ctx, res->nb_idx.reserve((size_t)total_nb * 4 + 16));
        ssg_launch_density<R>(Q, d_xyz, ctx->cell_start.as<uint32_t>(), res->perm.as<uint32_t>(), res->rho.as<R>(), 2, nullptr,
                              res->nb_ptr.as<unsigned long long>(), res->nb_idx.as<uint32_t>(), st);
    } else {
        SS_HIP(ctx, hipMemsetAsync(res->nb_ptr.p, 0, 8, st));
        SS_HIP(ctx, hipEventRecord(ctx->ev[3], st));
    }
    return SS_OK;
}

// ---- global (non-decomposed) strategy: reconstruct_surface_global (reconstruction.rs:65-194), kernels in ss_global.hip ----
template <class R>
ss_status reconstruct_global(ss_context* ctx, const typename TypesOf<R>::params* prm, const typename TypesOf<R>::grid& grid, const R* d_xyz, uint32_t n,
                             bool host_input, ss_result* res) {
    hipStream_t st = ctx->stream;
    ss_status s = SS_OK;
    result_grid<R>(res) = grid;
    memset(&result_subgrid<R>(res), 0, sizeof(typename TypesOf<R>::grid));
    res->is_f64 = sizeof(R) == 8;
    res->global_strategy = true;
    res->host_input = host_input;
    res->n_occupied_subdomains = res->n_subdomain_particles = 0;
    res->n_active = res->n_mc = 0;

    SSGlobT<R> Q;
    memset(&Q, 0, sizeof(Q));
    const R h = prm->compact_support_radius, cs = prm->cube_size;
    double npts_d = 1.0;
    for (int d = 0; d < 3; ++d) {
        if (grid.n_points[d] > 2000000000ll) return fail(ctx, SS_ERR_GRID_CONSTRUCTION, "too many grid points per dimension", SS_GRID_INDEX_TYPE_TOO_SMALL);
        Q.gmin[d] = grid.aabb_min[d];
        Q.np[d] = (int)grid.n_points[d];
        Q.nc[d] = (int)grid.n_cells[d];
        npts_d *= (double)grid.n_points[d];
    }
    if (npts_d >= 2.0e9)
        return fail(ctx, SS_ERR_UNSUPPORTED,
                    "global (non-decomposed) strategy: the dense level-set array of this build holds < 2e9 grid points; use the uniform-grid decomposition");
    Q.cs = cs;
    Q.h = h;
    Q.h2 = h * h;
    Q.sigma = R(8.0) / (h * h * h);
    Q.w0 = ss_kernel_evaluate(R(0.0), h, Q.sigma);
    {
        const R d2 = prm->particle_radius + prm->particle_radius;  // Volume::cube_particle (kernel.rs:28-30)
        Q.mass = d2 * d2 * d2 * prm->rest_density;
    }
    Q.threshold = prm->iso_surface_threshold;
    Q.n = n;
    // neighbourhood-search grid over grid.aabb() (neighborhood_search.rs:163-176)
    typename TypesOf<R>::grid sgrid;
    if (grid_from_aabb<R>(&sgrid, grid.aabb_min, grid.aabb_max, h) != 0)
        return fail(ctx, SS_ERR_UNKNOWN, "domain for neighborhood search has to be consistent and not degenerate (reference: assert)");
    double scell_d = 1.0;
    for (int d = 0; d < 3; ++d) {
        Q.smin[d] = sgrid.aabb_min[d];
        if (sgrid.n_cells[d] > 2000000000ll) return fail(ctx, SS_ERR_UNSUPPORTED, "search grid too large");
        Q.snc[d] = (int)sgrid.n_cells[d];
        scell_d *= (double)sgrid.n_cells[d];
    }
    if (scell_d >= 4.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "search grid too large for the dense cell table of this build");
    // SparseDensityMapGenerator::try_new (density_map.rs:582-640)
    const R half_real = ss_ceil(h / cs);
    Q.half_cells = (int)half_real;
    Q.supported = 2 * Q.half_cells + 2;
    const R radius = cs * half_real * (R(1.0) + ss_sqrt(std::numeric_limits<R>::epsilon()));
    Q.radius_sq = radius * radius;
    const R neg = -radius;
    bool degenerate = true, consistent = true;
    for (int d = 0; d < 3; ++d) {
        Q.amin[d] = grid.aabb_min[d] - neg;  // grow_uniformly(-radius), aabb.rs:257-260
        Q.amax[d] = grid.aabb_max[d] + neg;
        degenerate = degenerate && Q.amin[d] == Q.amax[d];
        consistent = consistent && Q.amin[d] <= Q.amax[d];
    }
    if (sizeof(R) == 4)
        res->Q32 = *reinterpret_cast<SSGlobT<float>*>(&Q);
    else
        res->Q64 = *reinterpret_cast<SSGlobT<double>*>(&Q);
    SS_HIP(ctx, hipEventRecord(ctx->ev[2], st));

    const size_t npts = (size_t)npts_d;
    // ---- neighbourhood search + densities ----
    s = global_search_and_densities<R>(ctx, Q, d_xyz, res);
    if (s != SS_OK) return s;
    SS_HIP(ctx, hipEventRecord(ctx->ev[4], st));

    // ---- sparse density map -> dense level-set array (density_map.rs:364-412) ----
    if (degenerate || !consistent)
        return fail(ctx, SS_ERR_DENSITY_MAP,
                    "the allowed domain of particles is inconsistent/degenerate (DensityMapError::InvalidDomain, density_map.rs:615-627)");
    const size_t n_chunks = ((size_t)n + SS_GCHUNK - 1) / SS_GCHUNK;
    const double tiles_d = std::ceil(Q.np[0] / 8.0) * std::ceil(Q.np[1] / 8.0) * std::ceil(Q.np[2] / 8.0);
    if (tiles_d * (double)n_chunks > 8.0e9)
        return fail(ctx, SS_ERR_UNSUPPORTED,
                    "global (non-decomposed) strategy: input too large for this build (level-set tiles x particle chunks > 8e9); "
                    "use the uniform-grid decomposition, which is the path optimised for large inputs");
    SS_HIP(ctx, res->G.reserve(npts * sizeof(R) + 16));
    SS_HIP(ctx, ctx->gboxes.reserve(n_chunks * 6 * sizeof(int) + 16));
    ssg_launch_chunk_boxes<R>(Q, d_xyz, ctx->gboxes.as<int>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[5], st));
    ssg_launch_levelset<R>(Q, d_xyz, res->rho.as<R>(), ctx->gboxes.as<int>(), res->G.as<R>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[6], st));

    // ---- marching cubes (narrow_band_extraction.rs, triangulation.rs) ----
    uint64_t nv = 0, nt = 0;
    s = global_marching_cubes<R>(ctx, Q, res, &nv, &nt);
    if (s != SS_OK) return s;
    res->n_vertices = nv;
    res->n_triangles = nt;
    ss_stats& S = res->stats;
    S.ms_total = ev_ms(ctx, 0, 9);
    S.ms_upload = host_input ? ev_ms(ctx, 0, 1) : 0.0;
    S.ms_aabb_grid = ev_ms(ctx, 1, 2);
    S.ms_decomposition = ev_ms(ctx, 2, 3);  // cell -> particle map of the neighbourhood search
    S.ms_density = ev_ms(ctx, 3, 4);
    S.ms_levelset_prepare = ev_ms(ctx, 4, 5);
    S.ms_levelset = ev_ms(ctx, 5, 6);
    S.ms_marching_cubes = ev_ms(ctx, 6, 7) + ev_ms(ctx, 8, 9);
    S.ms_stitching = ev_ms(ctx, 7, 8);
    S.n_particles = n;
    S.n_vertices = nv;
    S.n_triangles = nt;
    S.levelset_kernel_launches = 1;
    size_t held = 0;
    for (const DevBuf* b : {&ctx->xyz_in, &ctx->xyz_filt, &ctx->keys_a, &ctx->keys_b, &ctx->vals_a, &ctx->cell_count, &ctx->cell_start, &ctx->temp, &ctx->vcount,
                            &ctx->tcount, &ctx->gboxes, &ctx->nb_count, &ctx->nb_tmp, &res->rho, &res->perm, &res->inside8, &res->G, &res->masks, &res->vbase,
                            &res->tbase, &res->vertices, &res->vkeys, &res->tri32, &res->nb_ptr, &res->nb_idx})
        held += b->cap;
    S.bytes_device_peak = held;
    res->valid = true;
    res->phase = 2;
    return SS_OK;
}

template <class R>
ss_status phase_finish(ss_context* ctx, ss_result* res);

// Once per distinct h (f32 only): prove on the device that the reciprocal division used inside W is exact for this divisor
// (k_verify_fast_div); the density and splat kernels use their generic variants otherwise.
template <class R>
static ss_status ensure_fast_div(ss_context* ctx, R h, hipStream_t st) {
    if constexpr (sizeof(R) == 4) {
        if (ctx->fastdiv_h == h) return SS_OK;
        ctx->fastdiv_ok = false;
        if (h > R(1.0e-9) && h < R(1.0e15)) {
            uint32_t bad = 1;
            SS_HIP(ctx, ctx->fastdiv_scratch.reserve(64));
            SS_HIP(ctx, hipMemsetAsync(ctx->fastdiv_scratch.p, 0, 4, st));
            ss_launch_verify_fast_div(h, R(1.0) / h, ctx->fastdiv_scratch.as<uint32_t>(), st);
            SS_HIP(ctx, hipMemcpyAsync(&bad, ctx->fastdiv_scratch.p, 4, hipMemcpyDeviceToHost, st));
            SS_HIP(ctx, hipStreamSynchronize(st));
            ++ctx->host_waits;  // (once per distinct h and context)
            ctx->fastdiv_ok = (bad == 0);
        }
        ctx->fastdiv_h = h;
    }
    return SS_OK;
}

// Phase 1: staging, grid, binning, densities.  `shard` == nullptr: the whole domain (single process).
template <class R>
ss_status phase_begin(ss_context* ctx, const R* xyz, uint64_t n_in, const typename TypesOf<R>::params* prm, const typename TypesOf<R>::shard* shard, ss_result* res) {
    ss_status s = validate_params(ctx, prm, n_in);
    if (s != SS_OK) return s;
    if (shard && prm->has_particle_aabb) return fail(ctx, SS_ERR_UNSUPPORTED, "particle_aabb cannot be combined with a shard descriptor");
    res->phase = 0;
    SS_HIP(ctx, hipSetDevice(ctx->device));
    s = ensure_events(ctx);
    if (s != SS_OK) return s;
    hipStream_t st = ctx->stream;
    res->valid = false;
    reset_host_flags(res);
    memset(&res->stats, 0, sizeof(res->stats));
    res->n_input = n_in;
    res->n_vertices = res->n_triangles = 0;
    res->n_active = res->n_mc = 0;
    res->dbg_certified = nullptr;
    ++ctx->call_serial;  // (whatever an earlier result still points to in this context's scratch is stale from here on: ss_result_debug_certified)
    ctx->host_waits = 0;

    const bool host_input = n_in > 0 && xyz && !is_device_pointer(xyz);
    SS_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    const R* d_xyz = nullptr;
    uint32_t n = 0;
    s = stage_particles(ctx, xyz, n_in, prm, res, &d_xyz, &n);
    if (s != SS_OK) return s;
    res->n_particles = n;
    res->density_kernel_timed = false;
    SS_HIP(ctx, hipEventRecord(ctx->ev[1], st));

    // ---- grid set-up (lib.rs:409-417, reconstruction.rs:24-29) ----
    R pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
    if (shard) {
        // multi-GPU: the grid is that of the WHOLE job; the caller supplies the AABB of all particles
        for (int d = 0; d < 3; ++d) {
            pmin[d] = shard->domain_min[d];
            pmax[d] = shard->domain_max[d];
        }
    } else if (!prm->has_particle_aabb && n > 0) {
        s = compute_particle_aabb(ctx, d_xyz, n, pmin, pmax);
        if (s != SS_OK) return s;
    }
    typename TypesOf<R>::grid initial;
    int gerr = grid_for_reconstruction(prm, shard ? true : (n > 0), pmin, pmax, &initial);
    if (gerr) return fail(ctx, SS_ERR_GRID_CONSTRUCTION, "grid construction failed (uniform_grid.rs:147-169)", gerr);
    res->global_strategy = false;
    if (use_global_strategy<R>(prm, initial)) {
        if (shard)
            return fail(ctx, SS_ERR_UNSUPPORTED, "the multi-GPU shard extension needs the uniform-grid decomposition (decomposition=1, auto_disable=0)");
        return reconstruct_global<R>(ctx, prm, initial, d_xyz, n, host_input, res);
    }
    R mass = 0, margin = 0;
    initialize_subdomain_parameters(prm, &initial, &result_grid<R>(res), &result_subgrid<R>(res), &mass, &margin);
    for (int d = 0; d < 3; ++d)
        if (result_grid<R>(res).n_points[d] > 2000000000ll) return fail(ctx, SS_ERR_GRID_CONSTRUCTION, "too many grid points per dimension", SS_GRID_INDEX_TYPE_TOO_SMALL);
    SSDevT<R> P;
    s = make_device_params<R>(ctx, prm, result_grid<R>(res), result_subgrid<R>(res), mass, margin, n, shard, &P);
    if (s != SS_OK) return s;
    dev_params<R>(res) = P;
    res->is_f64 = sizeof(R) == 8;
    res->host_input = host_input;
    SS_HIP(ctx, hipEventRecord(ctx->ev[2], st));

    const size_t ncells = (size_t)P.kdim[0] * P.kdim[1] * P.kdim[2];

    // ---- K1: bin + sort (decomposition) ----
    s = ensure_mail(ctx);
    if (s != SS_OK) return s;
    SS_HIP(ctx, res->rho.reserve(ss_round16((size_t)n * sizeof(R) + 16)));
    SS_HIP(ctx, res->posvol.reserve((size_t)n * sizeof(ss_real4<R>) + 32));
    SS_HIP(ctx, res->posvol_by_index.reserve((size_t)n * sizeof(ss_real4<R>) + 32));
    SS_HIP(ctx, res->perm.reserve((size_t)n * 4 + 16));
    SS_HIP(ctx, ctx->cell_start.reserve((ncells + 1) * 4));
    SS_HIP(ctx, ctx->keys_a.reserve((size_t)n * 4 + 16));
    SS_HIP(ctx, ctx->keys_b.reserve((size_t)n * 4 + 16));
    SS_HIP(ctx, ctx->vals_a.reserve((size_t)n * 4 + 16));
    SS_HIP(ctx, ctx->pos_sorted.reserve((size_t)n * sizeof(ss_pos<R>) + 16));
    const size_t nsub = (size_t)P.ns[0] * P.ns[1] * P.ns[2];
    if (nsub >= SS_SCAN_MAX_N) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 - 8193 subdomains");
    SS_HIP(ctx, ctx->copy_offset.reserve(((size_t)n + 1) * 4));
    SS_HIP(ctx, ctx->sub_rank.reserve((nsub + 1) * 4));
    SS_HIP(ctx, ctx->occ_sub.reserve((nsub + 1) * 4 + 16));  // (at most every subdomain is occupied)
    // ---- K1: the splat-cell sort.  Independent of the densities; with SPLASH_K1_OVERLAP=1 the chain runs on a second stream beside the density kernel
    // (forked right before it, joined at the end of the phase) -- measured, no gain worth its complexity (ss_host.h), off by default.
    unsigned bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < ncells) ++bits;
    auto launch_k1 = [&](hipStream_t s1) -> ss_status {
        // zeroed words of the chain, one memset: a scan state, the sort's work words and the run starts of the cell table (0 = empty cell)
        const size_t words = (ss_scan_state_words(ncells + 1) + ss_radix_sort_work_words(n, bits) + (ncells + 1) + 64 + 3) & ~(size_t)3;  // (whole 16-byte units: ss_round16)
        SS_HIP(ctx, ctx->zeros_k1.reserve(words * 4));
        SS_HIP(ctx, hipMemsetAsync(ctx->zeros_k1.p, 0, words * 4, s1));
        ZeroTaker Z1;
        Z1.base = ctx->zeros_k1.as<uint32_t>();
        Z1.cap = words;
        uint32_t* st_cells = Z1.take(ss_scan_state_words(ncells + 1));
        uint32_t* sort_work = Z1.take(ss_radix_sort_work_words(n, bits));
        uint32_t* cell_first = Z1.take(ncells + 1);
        if (!cell_first) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
        const uint32_t* sorted_keys = ctx->keys_a.as<uint32_t>();
        if (n > 0) {
            // the buffers are assigned so that the sorted positions end in res->perm whatever the number of passes
            const bool odd = (((bits + 7u) / 8u) & 1u) != 0u;
            uint32_t* keys[2] = {ctx->keys_a.as<uint32_t>(), ctx->keys_b.as<uint32_t>()};
            uint32_t* vals[2] = {odd ? ctx->vals_a.as<uint32_t>() : res->perm.as<uint32_t>(), odd ? res->perm.as<uint32_t>() : ctx->vals_a.as<uint32_t>()};
            ss_launch_cell_keys(P, d_xyz, keys[0], (uint32_t*)nullptr, s1);
            int r = 0;
            ss_status s2 = sort_pairs(ctx, keys, vals, n, bits, true, sort_work, &r, s1);
            if (s2 != SS_OK) return s2;
            if (vals[r] != res->perm.as<uint32_t>()) return fail(ctx, SS_ERR_UNKNOWN, "internal error: sort result in an unexpected buffer");
            sorted_keys = keys[r];
        }
        ss_launch_sorted_gather_runs(P, n, d_xyz, res->perm.as<uint32_t>(), ctx->pos_sorted.as<ss_pos<R>>(), sorted_keys, (uint32_t)ncells, cell_first, (const uint32_t*)nullptr,
                                     (uint8_t*)nullptr, s1);
        ss_launch_cell_table_scan(cell_first, (uint32_t)ncells, ctx->cell_start.as<uint32_t>(), st_cells, s1);
        return SS_OK;
    };
    bool k1_done = false;
    if (ctx->overlap_k1 && !ctx->stream2) {
        if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            ctx->overlap_k1 = false;
        }
    }
    if (!ctx->overlap_k1) {  // one stream: K1 first, as in rounds 1-3
        SS_HIP(ctx, hipEventRecord(ctx->ev[22], st));
        s = launch_k1(st);
        if (s != SS_OK) return s;
        SS_HIP(ctx, hipEventRecord(ctx->ev[23], st));
        k1_done = true;
    }
    // zeroed words of the density preparation up to the first count the host waits for, ONE memset: two scan states and the subdomain flags
    ZeroTaker Z;
    s = reserve_zeros(ctx, ss_scan_state_words(n) + ss_scan_state_words(nsub) + (nsub + 1) + 64, &Z);
    if (s != SS_OK) return s;
    uint32_t* st_member = Z.take(ss_scan_state_words(n));
    uint32_t* st_sub = Z.take(ss_scan_state_words(nsub));
    uint32_t* sub_flag = Z.take(nsub + 1);
    if (!sub_flag) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
    SS_HIP(ctx, hipEventRecord(ctx->ev[3], st));

    // ---- K2: densities (per-subdomain particle copies, exactly the reference's organisation) ----
    {
        const double ctot_d = (double)P.sc[0] * P.sc[1] * P.sc[2];
        SS_HIP(ctx, hipMemsetAsync(res->rho.p, 0, ss_round16((size_t)n * sizeof(R) + 16), st));  // vec![R::zero(); n], dense_subdomains.rs:504
        // member counts -> copy offsets (the membership count as the scan's input), occupied subdomains -> ranks and list
        const SSMailSlot m_copies = mail_slot(ctx, 0), m_occ = mail_slot(ctx, 1);
        ss_launch_classify_scan(P, d_xyz, ctx->copy_offset.as<uint32_t>(), sub_flag, st_member, m_copies, st);
        ss_launch_flag_scan(sub_flag, (uint32_t)nsub, ctx->sub_rank.as<uint32_t>(), ctx->occ_sub.as<uint32_t>(), nullptr, st_sub, m_occ, st);
        unsigned long long v_copies = 0, v_occ = 0;
        s = mail_wait(ctx, m_copies, &v_copies);
        if (s != SS_OK) return s;
        s = mail_wait(ctx, m_occ, &v_occ, true);
        if (s != SS_OK) return s;
        if (v_copies >= (1ull << 30)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^30 - 1 (particle, subdomain) pairs in one call are not supported by this build");
        const uint32_t n_copies = (uint32_t)v_copies, n_occ = (uint32_t)v_occ;
        res->n_occupied_subdomains = n_occ;
        res->n_subdomain_particles = n_copies;
        if ((double)n_occ * ctot_d > 4.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "too many (subdomain, search cell) pairs for this build");
        const size_t ncells2 = (size_t)n_occ * (size_t)ctot_d;
        if (n_copies > 0) {
            SS_HIP(ctx, ctx->ckeys_a.reserve((size_t)n_copies * 4 + 16));
            SS_HIP(ctx, ctx->ckeys_b.reserve((size_t)n_copies * 4 + 16));
            SS_HIP(ctx, ctx->cvals_a.reserve((size_t)n_copies * 4 + 16));
            SS_HIP(ctx, ctx->cidx.reserve((size_t)n_copies * 4 + 16));
            SS_HIP(ctx, ctx->cpos.reserve(((size_t)n_copies + 16) * sizeof(ss_pos<R>)));  // + padding: k_density_sub reads whole chunks
            SS_HIP(ctx, ctx->cell_start2.reserve((ncells2 + 1) * 4));
            SS_HIP(ctx, ctx->own_flag.reserve(((size_t)n_copies + 16) * 5 + 128));  // the list of owned copies (u32), then their flags (u8)
            unsigned bits = 1;
            while (bits < 32 && ((size_t)1 << bits) < ncells2) ++bits;
            ZeroTaker Z2;  // (one memset again: scan states, the count of owned copies, the sort's work words, the run starts of the copies' cell table)
            s = reserve_zeros(ctx, ss_scan_state_words(ncells2 + 1) + ss_scan_state_words(n_copies) + ss_radix_sort_work_words(n_copies, bits) + (ncells2 + 1) + 64, &Z2);
            if (s != SS_OK) return s;
            uint32_t* st_cells2 = Z2.take(ss_scan_state_words(ncells2 + 1));
            uint32_t* st_owned = Z2.take(ss_scan_state_words(n_copies));
            uint32_t* n_owned_dev = Z2.take(4);
            uint32_t* sort_work2 = Z2.take(ss_radix_sort_work_words(n_copies, bits));
            uint32_t* cell_first2 = Z2.take(ncells2 + 1);
            if (!cell_first2) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
            const bool odd = (((bits + 7u) / 8u) & 1u) != 0u;
            uint32_t* keys[2] = {ctx->ckeys_a.as<uint32_t>(), ctx->ckeys_b.as<uint32_t>()};
            uint32_t* vals[2] = {odd ? ctx->cvals_a.as<uint32_t>() : ctx->cidx.as<uint32_t>(), odd ? ctx->cidx.as<uint32_t>() : ctx->cvals_a.as<uint32_t>()};
            ss_launch_emit_copies(P, d_xyz, ctx->copy_offset.as<uint32_t>(), ctx->sub_rank.as<uint32_t>(), keys[0], vals[0], st);
            int r = 0;
            s = sort_pairs(ctx, keys, vals, n_copies, bits, false, sort_work2, &r);
            if (s != SS_OK) return s;
            if (vals[r] != ctx->cidx.as<uint32_t>()) return fail(ctx, SS_ERR_UNKNOWN, "internal error: sort result in an unexpected buffer");
            const uint32_t* ckeys_sorted = keys[r];
            uint32_t* own_list = ctx->own_flag.as<uint32_t>();
            uint8_t* own_flags = reinterpret_cast<uint8_t*>(own_list + ((size_t)n_copies + 16));
            ss_launch_sorted_gather_runs(P, n_copies, d_xyz, ctx->cidx.as<uint32_t>(), ctx->cpos.as<ss_pos<R>>(), ckeys_sorted, (uint32_t)ncells2, cell_first2,
                                         ctx->occ_sub.as<uint32_t>(), own_flags, st);
            ss_launch_cell_table_scan(cell_first2, (uint32_t)ncells2, ctx->cell_start2.as<uint32_t>(), st_cells2, st);
            const bool want_nb = prm->global_neighborhood_list != 0;
            if (want_nb) {
                SS_HIP(ctx, ctx->nb_count.reserve(ss_round16(((size_t)n + 1) * 8)));
                SS_HIP(ctx, res->nb_ptr.reserve(ss_round16(((size_t)n + 1) * 8)));
                SS_HIP(ctx, hipMemsetAsync(ctx->nb_count.p, 0, ss_round16(((size_t)n + 1) * 8), st));
            }
            s = ensure_fast_div<R>(ctx, P.h, st);
            if (s != SS_OK) return s;
            // the copies whose density their subdomain computes (every particle has exactly one), compacted in cell order
            ss_launch_owned_scan(n_copies, own_flags, own_list, n_owned_dev, st_owned, st);
            const uint32_t n_owned_bound = n < n_copies ? n : n_copies;  // at most one owned copy per particle
            if (ctx->overlap_k1) {  // fork: the K1 chain on the second stream, beside the density kernel
                SS_HIP(ctx, hipEventRecord(ctx->ev[24], st));
                SS_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev[24], 0));
                SS_HIP(ctx, hipEventRecord(ctx->ev[22], ctx->stream2));
                s = launch_k1(ctx->stream2);
                if (s != SS_OK) return s;
                SS_HIP(ctx, hipEventRecord(ctx->ev[23], ctx->stream2));
                SS_HIP(ctx, hipEventRecord(ctx->ev[25], ctx->stream2));
                k1_done = true;
            }
            SS_HIP(ctx, hipEventRecord(ctx->ev[18], st));
            ss_launch_density_sub(P, n_copies, ctx->cpos.as<ss_pos<R>>(), ctx->cidx.as<uint32_t>(), ckeys_sorted,
                                  ctx->cell_start2.as<uint32_t>(), ctx->occ_sub.as<uint32_t>(), res->rho.as<R>(), want_nb ? 1 : 0,
                                  ctx->nb_count.as<uint32_t>(), nullptr, nullptr, sizeof(R) == 4 && ctx->fastdiv_ok, own_list, n_owned_dev, n_owned_bound, st);
            SS_HIP(ctx, hipEventRecord(ctx->ev[19], st));
            res->density_kernel_timed = true;
            if (want_nb) {
                // counts (u32, first n+1 entries) -> u64 -> exclusive scan = CSR row pointers
                SS_HIP(ctx, ctx->nb_tmp.reserve(((size_t)n + 1) * 8));
                ss_launch_widen(ctx->nb_count.as<uint32_t>(), (size_t)n + 1, ctx->nb_tmp.as<unsigned long long>(), st);
                s = exclusive_scan_u32<unsigned long long>(ctx, ctx->nb_tmp.as<unsigned long long>(), res->nb_ptr.as<unsigned long long>(), (size_t)n + 1);
                if (s != SS_OK) return s;
                unsigned long long total_nb = 0;
                SS_HIP(ctx, hipMemcpyAsync(&total_nb, res->nb_ptr.as<unsigned long long>() + n, 8, hipMemcpyDeviceToHost, st));
                SS_HIP(ctx, hipStreamSynchronize(st));
                ++ctx->host_waits;
                if (total_nb >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 neighbour entries");
                res->n_neighbors = total_nb;
                SS_HIP(ctx, res->nb_idx.reserve((size_t)total_nb * 4 + 16));
                ss_launch_density_sub(P, n_copies, ctx->cpos.as<ss_pos<R>>(), ctx->cidx.as<uint32_t>(), ckeys_sorted,
                                      ctx->cell_start2.as<uint32_t>(), ctx->occ_sub.as<uint32_t>(), res->rho.as<R>(), 2, nullptr,
                                      res->nb_ptr.as<unsigned long long>(), res->nb_idx.as<uint32_t>(), false, own_list, n_owned_dev, n_owned_bound, st);
            }
            res->has_neighbors = want_nb;
        } else {
            res->has_neighbors = prm->global_neighborhood_list != 0;
            res->n_neighbors = 0;
            if (res->has_neighbors) {
                SS_HIP(ctx, res->nb_ptr.reserve(ss_round16(((size_t)n + 1) * 8)));
                SS_HIP(ctx, hipMemsetAsync(res->nb_ptr.p, 0, ss_round16(((size_t)n + 1) * 8), st));
            }
        }
    }
    if (k1_done) {
        if (ctx->overlap_k1) SS_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[25], 0));  // join
    } else {  // (no particle has a copy: there was no density kernel to run beside)
        SS_HIP(ctx, hipEventRecord(ctx->ev[22], st));
        s = launch_k1(st);
        if (s != SS_OK) return s;
        SS_HIP(ctx, hipEventRecord(ctx->ev[23], st));
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[4], st));
    res->phase = 1;
    return SS_OK;
}

// Phase 2: level set, marching cubes, numbering.  Uses res->rho as it is on the device NOW (a
// multi-GPU host may have filled in the densities of halo particles between the phases).
template <class R>
ss_status phase_finish(ss_context* ctx, ss_result* res) {
    if (res->phase != 1) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "ss_shard_finish without a preceding successful ss_shard_begin_f32");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ss_status s = SS_OK;
    const SSDevT<R> P = dev_params<R>(res);
    const uint32_t n = P.n;
    const bool host_input = res->host_input;
    const size_t ncells = (size_t)P.kdim[0] * P.kdim[1] * P.kdim[2];
    const size_t nblocks = (size_t)P.bt_dim[0] * P.bt_dim[1] * P.bt_dim[2];  // (the table window: the whole grid unless this is a shard)
    s = ensure_mail(ctx);
    if (s != SS_OK) return s;
    SS_HIP(ctx, hipEventRecord(ctx->ev[10], st));
    ss_launch_make_posvol(P, ctx->pos_sorted.as<ss_pos<R>>(), res->perm.as<uint32_t>(), res->rho.as<R>(), res->posvol.as<ss_real4<R>>(), res->posvol_by_index.as<ss_real4<R>>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[11], st));

    // ---- K3 prepare: active level-set blocks ----
    if (nblocks >= SS_SCAN_MAX_N) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 - 8193 level-set blocks");
    SS_HIP(ctx, res->block_slot.reserve(nblocks * 4 + 16));
    SS_HIP(ctx, res->mc_slot.reserve(nblocks * 4 + 16));
    ZeroTaker ZB;  // the block flags and the state of the scan over them (two states: the scan may be repeated), one memset
    s = reserve_zeros(ctx, (nblocks + 1) + 2 * ss_scan_state_words(nblocks) + 64, &ZB);
    if (s != SS_OK) return s;
    uint32_t* block_flag = ZB.take(nblocks + 1);
    if (n > 0) ss_launch_mark_blocks(P, ctx->cell_start.as<uint32_t>(), (uint32_t)ncells, block_flag, st);
    // flags -> slot table, block coordinates (list order = table order); the lists are filled up to their capacity: a call that has more active
    // blocks than any before it repeats the scan with larger lists
    uint32_t n_active = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (ctx->cap_active == 0) ctx->cap_active = (uint32_t)std::min<size_t>(nblocks, (size_t)1 << 16);
        SS_HIP(ctx, res->active_xyz.reserve((size_t)ctx->cap_active * 12 + 16));
        const SSMailSlot m_active = mail_slot(ctx, 2);
        uint32_t* st_active = ZB.take(ss_scan_state_words(nblocks));
        if (!block_flag || !st_active) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
        ss_launch_active_blocks_scan(P, block_flag, (uint32_t)nblocks, ctx->cap_active, (uint32_t*)nullptr, res->block_slot.as<uint32_t>(),
                                     res->active_xyz.as<uint32_t>(), st_active, m_active, st);
        unsigned long long v = 0;
        s = mail_wait(ctx, m_active, &v);
        if (s != SS_OK) return s;
        n_active = (uint32_t)v;
        if (n_active <= ctx->cap_active) break;
        ctx->cap_active = (uint32_t)std::min<size_t>(nblocks, (size_t)n_active + n_active / 4 + 1024);
    }
    if (n_active > ctx->cap_active) return fail(ctx, SS_ERR_UNKNOWN, "internal error: the list of active blocks overflowed twice");
    res->n_active = n_active;
    SS_HIP(ctx, res->G.reserve((size_t)n_active * SS_BLOCK_POINTS * sizeof(R) + 16));
    SS_HIP(ctx, res->blk_minmax.reserve((size_t)n_active * 2 * sizeof(R) + 16));

    // ---- K3: level-set splat (ss_kernels.hip) ----
    {
        const bool checked_now = sizeof(R) == 4 && ctx->fastdiv_h != (float)P.h;
        ss_status fs = ensure_fast_div<R>(ctx, P.h, st);  // normally done before the densities already
        if (fs != SS_OK) return fs;
        (void)checked_now;
    }
    const R prm_threshold = P.threshold;
    SSDevT<R> PK = P;
    if (sizeof(R) == 4) {
        const bool lean_ok = P.h > R(1.0e-9) && P.h < R(1.0e15);  // range in which the lean exact sqrt needs no scaling
        if (P.arith == SS_ARITH_GENERIC && ctx->fastdiv_ok) PK.arith = SS_ARITH_FAST;
        if (P.arith == SS_ARITH_SIMD && lean_ok) PK.arith = SS_ARITH_SIMD_LEAN;
    }
    SS_HIP(ctx, ctx->splat_overflow.reserve(((size_t)n_active + 1) * 4 * 4 + 64));
    uint32_t* lg_flag = ctx->splat_overflow.as<uint32_t>();
    uint32_t* lg_rank = lg_flag + ((size_t)n_active + 1);
    uint32_t* lg_list = lg_rank + ((size_t)n_active + 1);
    SS_HIP(ctx, ctx->splat_counts.reserve(((size_t)n_active + 1) * 4));
    uint64_t n_reserved = 0;
    // Sub-blocks that a cheap lower bound certifies to lie inside the fluid are not evaluated in full (ss_kernels.hip,
    // splat_accumulate_block_wave).  Tiny jobs (< 1 k active blocks) skip the scheme: its extra launches cost more than it saves there.
    // ... and so do workloads where the previous call certified too few sub-blocks to pay for the classification pass (break-even:
    // 35 % of the sub-blocks); such a workload is probed again every 16th call.
    uint64_t early_key = (uint64_t)n * 0x9E3779B97F4A7C15ull;
    {
        uint64_t hb = 0, cb = 0;
        memcpy(&hb, &P.h, sizeof(R));
        memcpy(&cb, &P.cs, sizeof(R));
        early_key ^= hb * 0xC2B2AE3D27D4EB4Full ^ (cb << 1);
    }
    if (ctx->early_key != early_key) {
        ctx->early_key = early_key;
        ctx->early_enabled = true;
        ctx->early_skipped = 0;
    }
    bool probe = ctx->early_enabled;
    if (!probe && ++ctx->early_skipped >= 16) {
        probe = true;
        ctx->early_skipped = 0;
    }
    const bool full_ls = ctx->full_levelset || !(prm_threshold > R(0.0)) || ctx->two_pass == 0 || (ctx->two_pass < 0 && (n_active < 1024u || !probe));
    SS_HIP(ctx, ctx->splat_trunc.reserve(((size_t)n_active + 2) * (8 + 8 * 4) + 64));
    unsigned long long* face_bits = ctx->splat_trunc.as<unsigned long long>();  // per block: faces of its sub-blocks with points outside the surface
    uint32_t* tr_flag = (uint32_t*)(face_bits + ((size_t)n_active + 2));        // per block: mask of the certified sub-blocks
    uint32_t* rd_flag = tr_flag + ((size_t)n_active + 1);         // ... and marching cubes will read it
    uint32_t* rd_unused = rd_flag + ((size_t)n_active + 1);
    uint32_t* rd_list = rd_unused + ((size_t)n_active + 1);
    uint32_t* big = rd_list + ((size_t)n_active + 1);            // blocks with more candidates than a wave holds (count, list): the arena path
    uint32_t* exact_list = big + ((size_t)n_active + 2);         // over-dense blocks with sub-blocks left to evaluate after k_splat_certify_big (count, list)
    // over-dense blocks of an f32 job: certificates straight from the cells first, tiles only for the blocks somebody reads (ss_kernels.hip)
    const bool certify_big = sizeof(R) == 4 && !full_ls;
    // zeroed words of the rest of this phase: the statistics counters, the states of the remaining scans, the length of the redo list
    const size_t mc_cap_bound = std::min<size_t>(nblocks, (size_t)8 * (size_t)n_active);
    ZeroTaker Z;
    // (the state of a scan whose length is not known yet is reserved with the monotone bound: ss_scan_state_words itself is not monotone)
    s = reserve_zeros(ctx, 16 + 3 * 64 * 2 + 6 * ss_scan_state_words((size_t)n_active + 1) + 2 * ss_scan_state_words(nblocks) + 2 * ss_scan_state_words_bound(mc_cap_bound + 1) + 2 * ((size_t)n_active + 8) + 64, &Z);
    if (s != SS_OK) return s;
    uint32_t* big_flag = Z.take((size_t)n_active + 2);   // flags of the blocks with more candidates than a wave holds (set by k_splat_fused, compacted into big[])
    uint32_t* need_mask = Z.take((size_t)n_active + 2);  // over-dense blocks: the sub-blocks k_splat_certify_big left to evaluate (0 for every other block)
    uint32_t* st_big1 = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* st_big2 = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* st_exact = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* d_counters = Z.take(3 * 64 * 2);  // u64[3][64]: tile entries, blocks left truncated, certified sub-blocks (64 copies each, k_select_redo)
    uint32_t* d_err = Z.take(4);                 // error flags
    uint32_t* st_tiles = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* st_large = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* st_redo = Z.take(ss_scan_state_words((size_t)n_active + 1));
    uint32_t* n_redo_dev = Z.take(4);
    uint32_t* n_large_dev = Z.take(4);
    if (!n_large_dev) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
    uint32_t n_big = 0;
    SS_HIP(ctx, hipEventRecord(ctx->ev[5], st));  // (= event 12 of the timers below: one record per point of the stream)
    if (n_active) {
        // first pass, gather and accumulate in one kernel: the tiles of ordinary blocks stay in LDS
        {   // the rows of splat cells a block scans (the same for every block): made once per parameter set
            const ss_context::RowTabKey key = {PK.sn1, PK.sk, PK.kdim[1], PK.kdim[2], (int)sizeof(R), PK.so, PK.se, PK.srho};
            const size_t need = (size_t)PK.sn1 * (size_t)PK.sn1 * 8 + 64;
            if (memcmp(&ctx->rowtab_key, &key, sizeof(key)) != 0 || ctx->splat_rowtab.cap < need) {
                SS_HIP(ctx, ctx->splat_rowtab.reserve(need));
                ss_launch_splat_row_table(PK, ctx->splat_rowtab.as<uint2>(), st);
                ctx->rowtab_key = key;
            }
        }
        ss_launch_splat_fused(PK, res->posvol.as<ss_real4<R>>(), res->perm.as<uint32_t>(), ctx->cell_start.as<uint32_t>(), ctx->splat_rowtab.as<uint2>(), res->active_xyz.as<uint32_t>(), n_active,
                              res->G.as<R>(), res->blk_minmax.as<ss_real2<R>>(), tr_flag, full_ls, nullptr, nullptr, nullptr, face_bits, ctx->splat_counts.as<uint32_t>(), big_flag, st);
        const SSMailSlot m_big = mail_slot(ctx, 3);
        ss_launch_flag_scan(big_flag, n_active, nullptr, big + 1, big, st_big1, m_big, st);  // flags -> (count, list in block order); the count goes to the host
        unsigned long long v = 0;
        s = mail_wait(ctx, m_big, &v);
        if (s != SS_OK) return s;
        n_big = (uint32_t)v;
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[16], st));
    if (n_big) {
        if constexpr (sizeof(R) == 4) {
            if (certify_big) {
                ss_launch_splat_certify_big(PK, res->posvol.as<ss_real4<float>>(), ctx->cell_start.as<uint32_t>(), res->active_xyz.as<uint32_t>(), n_active,
                                            res->block_slot.as<uint32_t>(), ctx->splat_counts.as<uint32_t>(), res->blk_minmax.as<ss_real2<float>>(), tr_flag, face_bits, need_mask, st);
                ss_launch_flag_scan(need_mask, n_active, nullptr, exact_list + 1, exact_list, st_exact, SSMailSlot{}, st);
            }
        }
        // over-dense blocks: bounds -> offsets -> tile arena -> gather (-> ordered list of the very large ones) -> workgroup per block
        SS_HIP(ctx, ctx->splat_off.reserve(((size_t)n_active + 2) * 8));
        SS_HIP(ctx, ctx->splat_bound.reserve(((size_t)n_active + 1) * 4));
        ss_launch_splat_bounds(PK, ctx->cell_start.as<uint32_t>(), res->active_xyz.as<uint32_t>(), n_active, ctx->splat_counts.as<uint32_t>(), ctx->splat_bound.as<uint32_t>(), st);
        const SSMailSlot m_arena = mail_slot(ctx, 4);
        ss_launch_tile_offsets_scan(ctx->splat_bound.as<uint32_t>(), n_active, ctx->splat_off.as<unsigned long long>(), st_tiles, m_arena, st);  // 64 bits: an arena can hold > 2^32 entries
        unsigned long long h_total = 0;
        s = mail_wait(ctx, m_arena, &h_total);
        if (s != SS_OK) return s;
        n_reserved = h_total;
        SS_HIP(ctx, ctx->splat_tiles.reserve((size_t)n_reserved * sizeof(ss_real4<R>) + 64));
        SS_HIP(ctx, ctx->splat_tile_idx.reserve((size_t)n_reserved * 4 + 64));
        ss_launch_splat_gather(PK, res->posvol.as<ss_real4<R>>(), res->perm.as<uint32_t>(), ctx->cell_start.as<uint32_t>(), res->active_xyz.as<uint32_t>(), n_active,
                               ctx->splat_off.as<unsigned long long>(), ctx->splat_tiles.as<ss_real4<R>>(), ctx->splat_tile_idx.as<uint32_t>(), ctx->splat_counts.as<uint32_t>(), lg_flag, st);
        ss_launch_posvol_by_index<R>(n, res->posvol.as<ss_real4<R>>(), res->perm.as<uint32_t>(), res->posvol_by_index.as<ss_real4<R>>(), st);  // (only calls with over-dense blocks need it)
        // very large tiles: flags -> ordered list on the device; the workgroup-level gather reads its length there
        ss_launch_flag_scan(lg_flag, n_active, nullptr, lg_list, n_large_dev, st_large, SSMailSlot{}, st);
        ss_launch_splat_gather_large(PK, res->posvol.as<ss_real4<R>>(), res->posvol_by_index.as<ss_real4<R>>(), res->perm.as<uint32_t>(), ctx->cell_start.as<uint32_t>(),
                                     res->active_xyz.as<uint32_t>(), lg_list, n_large_dev, ctx->splat_counts.as<uint32_t>(), ctx->splat_off.as<unsigned long long>(),
                                     ctx->splat_tiles.as<ss_real4<R>>(), ctx->splat_tile_idx.as<uint32_t>(), st);
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[17], st));
    if (n_big)
        ss_launch_splat_accumulate_big(PK, ctx->splat_tiles.as<ss_real4<R>>(), ctx->splat_tile_idx.as<uint32_t>(), ctx->splat_off.as<unsigned long long>(), ctx->splat_counts.as<uint32_t>(),
                                       res->active_xyz.as<uint32_t>(), res->G.as<R>(), res->blk_minmax.as<ss_real2<R>>(), tr_flag, full_ls, false, certify_big,
                                       certify_big ? need_mask : nullptr, face_bits, certify_big ? exact_list : big, d_err, st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[13], st));  // (= event 14)

    // second pass of the splat: certified sub-blocks with a face neighbour outside the surface are completed (list and count stay on the device);
    // the statistics of the first pass are taken by the same kernel
    if (n_active)
        ss_launch_select_redo(P, res->active_xyz.as<uint32_t>(), n_active, res->block_slot.as<uint32_t>(), full_ls ? nullptr : tr_flag, face_bits, rd_flag,
                              ctx->splat_counts.as<uint32_t>(), reinterpret_cast<unsigned long long*>(d_counters), big_flag, st);
    if (n_active && !full_ls) {
        ss_launch_flag_scan(rd_flag, n_active, nullptr, rd_list, n_redo_dev, st_redo, SSMailSlot{}, st);
        ss_launch_splat_fused(PK, res->posvol.as<ss_real4<R>>(), res->perm.as<uint32_t>(), ctx->cell_start.as<uint32_t>(), ctx->splat_rowtab.as<uint2>(), res->active_xyz.as<uint32_t>(), n_active,
                              res->G.as<R>(), res->blk_minmax.as<ss_real2<R>>(), tr_flag, true, rd_list, n_redo_dev, rd_flag, face_bits, ctx->splat_counts.as<uint32_t>(), big_flag, st);
        if (n_big) {  // (the list kernel flagged the large blocks among the selected ones again)
            ss_launch_flag_scan(big_flag, n_active, nullptr, big + 1, big, st_big2, SSMailSlot{}, st);
            ss_launch_splat_accumulate_big(PK, ctx->splat_tiles.as<ss_real4<R>>(), ctx->splat_tile_idx.as<uint32_t>(), ctx->splat_off.as<unsigned long long>(),
                                           ctx->splat_counts.as<uint32_t>(), res->active_xyz.as<uint32_t>(), res->G.as<R>(), res->blk_minmax.as<ss_real2<R>>(), tr_flag, true, true, false,
                                           rd_flag, face_bits, big, d_err, st);
        }
    }
    SS_HIP(ctx, hipEventRecord(ctx->ev[15], st));  // (= event 6)

    // ---- K4 prepare: MC blocks = blocks whose 2x2x2 level-set neighbourhood straddles the threshold (the flag is the scan's input) ----
    uint32_t n_mc = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (ctx->cap_mc == 0) ctx->cap_mc = (uint32_t)std::min<size_t>(mc_cap_bound, (size_t)1 << 16);
        if ((size_t)ctx->cap_mc > mc_cap_bound) ctx->cap_mc = (uint32_t)mc_cap_bound;
        SS_HIP(ctx, res->mc_xyz.reserve((size_t)ctx->cap_mc * 12 + 16));
        const SSMailSlot m_mc = mail_slot(ctx, 5);
        uint32_t* st_mc = Z.take(ss_scan_state_words(nblocks));
        if (!st_mc) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
        ss_launch_mc_blocks_scan(P, res->block_slot.as<uint32_t>(), res->blk_minmax.as<ss_real2<R>>(), (uint32_t)nblocks, ctx->cap_mc, (uint32_t*)nullptr, res->mc_slot.as<uint32_t>(),
                                 res->mc_xyz.as<uint32_t>(), st_mc, m_mc, st);
        unsigned long long v = 0;
        s = mail_wait(ctx, m_mc, &v);
        if (s != SS_OK) return s;
        n_mc = (uint32_t)v;
        if (n_mc <= ctx->cap_mc) break;
        ctx->cap_mc = (uint32_t)std::min<size_t>(mc_cap_bound, (size_t)n_mc + n_mc / 4 + 1024);
    }
    if (n_mc > ctx->cap_mc) return fail(ctx, SS_ERR_UNKNOWN, "internal error: the list of marching-cubes blocks overflowed twice");
    res->n_mc = n_mc;

    // ---- K4: MC classification + counts ----
    SS_HIP(ctx, res->masks.reserve((size_t)n_mc * 24 * 8 + 16));
    SS_HIP(ctx, ctx->vcount.reserve(((size_t)n_mc + 1) * 4));
    SS_HIP(ctx, ctx->tcount.reserve(((size_t)n_mc + 1) * 4));
    SS_HIP(ctx, res->vbase.reserve(((size_t)n_mc + 2) * 4));
    SS_HIP(ctx, res->tbase.reserve(((size_t)n_mc + 2) * 4));
    SS_HIP(ctx, ctx->mc_nb.reserve((size_t)n_mc * SS_MC_REC * 4 + 64));
    ss_launch_mc_neighbours(P, res->mc_xyz.as<uint32_t>(), n_mc, res->block_slot.as<uint32_t>(), res->mc_slot.as<uint32_t>(), full_ls ? nullptr : tr_flag, ctx->mc_nb.as<uint32_t>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[20], st));
    ss_launch_mc_count(P, res->G.as<R>(), ctx->mc_nb.as<uint32_t>(), res->mc_xyz.as<uint32_t>(), n_mc, res->masks.as<unsigned long long>(),
                       ctx->vcount.as<uint32_t>(), ctx->tcount.as<uint32_t>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[21], st));  // (= event 7)
    // ---- "stitching": global numbering by prefix sums (vertex and triangle counts in one scan) ----
    const SSMailSlot m_tot = mail_slot(ctx, 6), m_stat0 = mail_slot(ctx, 7), m_stat1 = mail_slot(ctx, 8), m_stat2 = mail_slot(ctx, 9), m_stat3 = mail_slot(ctx, 10);
    // one packed 31 + 31 bit scan while the worst case of the totals fits; two 64-bit scans for larger jobs (ss_kernels.hip, SSMcCountsIn)
    const bool split_offsets = ctx->split_mc_offsets || (uint64_t)n_mc * SS_MC_MAX_TRI_PER_BLOCK >= (1ull << 31);
    const SSMailSlot m_tot2 = split_offsets ? mail_slot(ctx, 16) : SSMailSlot{};
    uint32_t* st_off = Z.take(ss_scan_state_words((size_t)n_mc + 1));
    uint32_t* st_off2 = split_offsets ? Z.take(ss_scan_state_words((size_t)n_mc + 1)) : nullptr;
    if (!st_off || (split_offsets && !st_off2)) return fail(ctx, SS_ERR_UNKNOWN, "internal error: zero region too small");
    ss_launch_mc_offsets_scan(ctx->vcount.as<uint32_t>(), ctx->tcount.as<uint32_t>(), n_mc, res->vbase.as<uint32_t>(), res->tbase.as<uint32_t>(), st_off, st_off2, m_tot, m_tot2, st);
    ss_launch_publish_stats(reinterpret_cast<const unsigned long long*>(d_counters), n_redo_dev, n_large_dev, d_err, m_stat0, m_stat1, m_stat2, m_stat3, st);
    unsigned long long v_tot = 0, n_cand = 0, n_trunc_left = 0, n_cert_waves = 0, v_misc = 0;
    s = mail_wait(ctx, m_tot, &v_tot);
    if (s != SS_OK) return s;
    s = mail_wait(ctx, m_stat0, &n_cand, true);
    if (s != SS_OK) return s;
    s = mail_wait(ctx, m_stat1, &n_trunc_left, true);
    if (s != SS_OK) return s;
    s = mail_wait(ctx, m_stat2, &n_cert_waves, true);
    if (s != SS_OK) return s;
    s = mail_wait(ctx, m_stat3, &v_misc, true);  // n_redo | n_large << 24 | err << 56 ... see k_publish_stats
    if (s != SS_OK) return s;
    const uint32_t n_redo = (uint32_t)(v_misc & 0xFFFFFFFull), n_large = (uint32_t)((v_misc >> 28) & 0xFFFFFFFull), h_err = (uint32_t)(v_misc >> 56);
    uint64_t nv = v_tot & 0x7FFFFFFFull, nt = v_tot >> 31;
    if (split_offsets) {
        unsigned long long v_tot2 = 0;
        s = mail_wait(ctx, m_tot2, &v_tot2, true);
        if (s != SS_OK) return s;
        nv = v_tot;
        nt = v_tot2;
        if (nv >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 - 1 vertices in one call are not supported by this build");
    }
    if (h_err) return fail(ctx, SS_ERR_UNKNOWN, "internal error: a level-set block without a tile was asked for values (k_big_tile_select)");
    if (nt * 3 >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32/3 triangles in one call are not supported by this build");
    SS_HIP(ctx, res->vertices.reserve(nv * 3 * sizeof(R) + 16));
    SS_HIP(ctx, res->vkeys.reserve(nv * 8 + 16));
    SS_HIP(ctx, res->tri32.reserve(nt * 12 + 16));
    SS_HIP(ctx, hipEventRecord(ctx->ev[8], st));
    // ---- K5: emission ----
    ss_launch_mc_emit(P, res->G.as<R>(), ctx->mc_nb.as<uint32_t>(), res->mc_xyz.as<uint32_t>(), res->mc_slot.as<uint32_t>(), n_mc,
                      res->masks.as<unsigned long long>(), res->vbase.as<uint32_t>(), res->tbase.as<uint32_t>(), res->vertices.as<R>(),
                      res->vkeys.as<unsigned long long>(), res->tri32.as<uint32_t>(), st);
    SS_HIP(ctx, hipEventRecord(ctx->ev[9], st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    ++ctx->host_waits;  // the final drain
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(ctx, SS_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(e));
    }

    res->n_vertices = nv;
    res->n_triangles = nt;
    res->dbg_certified = full_ls ? nullptr : tr_flag;
    res->dbg_serial = ctx->call_serial;
    ss_stats& S = res->stats;
    S.ms_total = ev_ms(ctx, 0, 4) + ev_ms(ctx, 10, 9);  // both phases (excludes what the host does between them)
    S.ms_upload = host_input ? ev_ms(ctx, 0, 1) : 0.0;
    S.ms_aabb_grid = ev_ms(ctx, 1, 2);
    S.ms_decomposition = ev_ms(ctx, 22, 23);  // the K1 chain (on the second stream it runs beside the density kernel: its time is then part of ms_density's interval as well)
    S.ms_density = ev_ms(ctx, 3, 4) + ev_ms(ctx, 10, 11);
    S.ms_levelset_prepare = ev_ms(ctx, 11, 5);
    S.ms_levelset = ev_ms(ctx, 5, 15);
    S.ms_levelset_gather = ev_ms(ctx, 16, 17);  // the over-dense blocks' certificates and the arena path (0 without such blocks)
    S.ms_levelset_accumulate = ev_ms(ctx, 5, 13) - ev_ms(ctx, 16, 17) + ev_ms(ctx, 13, 15);  // both passes of the splat kernels (the second incl. its block selection)
    S.ms_marching_cubes = ev_ms(ctx, 15, 21) + ev_ms(ctx, 8, 9);
    S.ms_stitching = ev_ms(ctx, 21, 8);
    S.n_particles = n;
    S.n_vertices = nv;
    S.n_triangles = nt;
    S.n_active_blocks = n_active;
    S.n_block_candidates = n_cand;
    S.n_large_tile_blocks = n_large;
    S.fast_div_verified = ctx->fastdiv_ok ? 1 : 0;
    S.arith_mode = (uint64_t)PK.arith;
    S.bytes_tile_arena = (uint64_t)n_cand * sizeof(ss_real4<R>);
    S.bytes_tile_arena_reserved = (uint64_t)n_reserved * sizeof(ss_real4<R>);
    if (n_active && !full_ls) {
        const double frac = (double)n_cert_waves / (8.0 * (double)n_active);
        ctx->early_enabled = frac > (ctx->early_enabled ? 0.30 : 0.35);
    }
    S.ms_levelset_accumulate_pass2 = ev_ms(ctx, 13, 15);
    S.n_mc_blocks = n_mc;
    S.ms_density_kernel = res->density_kernel_timed ? ev_ms(ctx, 18, 19) : 0.0;
    S.ms_mc_count = ev_ms(ctx, 20, 21);
    S.ms_mc_emit = ev_ms(ctx, 8, 9);
    S.n_host_waits = ctx->host_waits;
    S.n_certified_subblocks = n_cert_waves;
    S.n_truncated_blocks = n_trunc_left;
    S.n_completed_blocks = n_redo;
    S.levelset_kernel_launches = n_active ? 1 : 0;
    size_t held = 0;
    for (const DevBuf* b : {&ctx->xyz_in, &ctx->xyz_filt, &ctx->flags32, &ctx->offsets, &ctx->keys_a, &ctx->keys_b, &ctx->vals_a, &ctx->cell_count,
                            &ctx->cell_start, &ctx->pos_sorted, &ctx->temp, &ctx->vcount, &ctx->tcount, &ctx->copy_offset, &ctx->ckeys_a, &ctx->ckeys_b, &ctx->cvals_a, &ctx->cidx,
                            &ctx->cpos, &ctx->cell_start2, &res->rho, &res->posvol, &ctx->mc_nb, &ctx->splat_tile_idx, &ctx->splat_tiles, &ctx->splat_counts, &ctx->splat_off, &ctx->splat_bound, &ctx->splat_overflow, &res->posvol_by_index, &res->perm, &res->inside8, &res->G, &res->block_slot,
                            &res->mc_slot, &res->masks, &res->vbase, &res->tbase, &res->vertices, &res->vkeys,
                            &res->tri32, &ctx->splat_trunc, &ctx->own_flag, &ctx->sub_rank, &ctx->occ_sub, &res->blk_minmax, &res->active_xyz, &res->mc_xyz, &ctx->zeros, &ctx->zeros_k1, &ctx->sort_work,
                            &ctx->aabb_partial, &ctx->fastdiv_scratch, &ctx->nb_count, &ctx->nb_tmp, &res->nb_ptr, &res->nb_idx})
        held += b->cap;
    S.bytes_device_peak = held;
    res->valid = true;
    res->phase = 2;
    return SS_OK;
}

template <class R>
ss_status reconstruct_impl(ss_context* ctx, const R* xyz, uint64_t n_in, const typename TypesOf<R>::params* prm, ss_result* res) {
    ss_status s = phase_begin<R>(ctx, xyz, n_in, prm, nullptr, res);
    if (s != SS_OK) return s;
    if (res->phase == 2) return SS_OK;  // the global strategy completes in one go
    return phase_finish<R>(ctx, res);
}

// ---- stand-alone entry points on the global strategy's stages ----
// marching_cubes::triangulate_density_map on a dense value array (marching_cubes.rs:100-127; pysplashsurf.marching_cubes)
template <class R>
ss_status marching_cubes_impl(ss_context* ctx, const R* values, const int64_t n_points[3], R threshold, R cube_size, const R translation[3], ss_result* res) {
    if (!ctx || !res || !n_points) return SS_ERR_INVALID_ARGUMENT;
    if (res->ctx != ctx) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context");
    ctx->err.clear();
    ctx->err_detail = 0;
    if (!(cube_size > R(0.0))) return fail(ctx, SS_ERR_GRID_CONSTRUCTION, "cube size must be positive (uniform_grid.rs:147-169)", SS_GRID_INVALID_CELL_SIZE);
    double npts_d = 1.0;
    for (int d = 0; d < 3; ++d) {
        if (n_points[d] < 2 || n_points[d] > 2000000000ll) return fail(ctx, SS_ERR_GRID_CONSTRUCTION, "every dimension of the value array needs at least 2 points");
        npts_d *= (double)n_points[d];
    }
    if (npts_d >= 2.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "value array too large for this build (< 2e9 points)");
    if (!values) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "values pointer is null");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ss_status s = ensure_events(ctx);
    if (s != SS_OK) return s;
    hipStream_t st = ctx->stream;
    res->valid = false;
    res->phase = 0;
    reset_host_flags(res);
    memset(&res->stats, 0, sizeof(res->stats));
    res->n_input = res->n_particles = 0;
    res->n_vertices = res->n_triangles = 0;
    res->has_inside = false;
    res->has_neighbors = false;
    res->n_neighbors = 0;
    res->is_f64 = sizeof(R) == 8;
    res->global_strategy = true;
    res->host_input = !is_device_pointer(values);
    typename TypesOf<R>::grid& g = result_grid<R>(res);
    R mn[3] = {R(0.0), R(0.0), R(0.0)};
    int64_t nc[3];
    for (int d = 0; d < 3; ++d) {
        if (translation) mn[d] = translation[d];
        nc[d] = n_points[d] - 1;
    }
    grid_new<R>(&g, mn, nc, cube_size);  // UniformGrid::new (uniform_grid.rs:203-232): min = translation, no alignment
    memset(&result_subgrid<R>(res), 0, sizeof(typename TypesOf<R>::grid));
    SSGlobT<R> Q;
    memset(&Q, 0, sizeof(Q));
    for (int d = 0; d < 3; ++d) {
        Q.gmin[d] = g.aabb_min[d];
        Q.np[d] = (int)n_points[d];
        Q.nc[d] = (int)nc[d];
    }
    Q.cs = cube_size;
    Q.threshold = threshold;
    if (sizeof(R) == 4)
        res->Q32 = *reinterpret_cast<SSGlobT<float>*>(&Q);
    else
        res->Q64 = *reinterpret_cast<SSGlobT<double>*>(&Q);
    const size_t npts = (size_t)npts_d;
    SS_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    SS_HIP(ctx, res->G.reserve(npts * sizeof(R) + 16));
    SS_HIP(ctx, hipMemcpyAsync(res->G.p, values, npts * sizeof(R), hipMemcpyDefault, st));
    SS_HIP(ctx, ctx->counter.reserve(64));
    SS_HIP(ctx, hipMemsetAsync(ctx->counter.p, 0, 64, st));
    SS_HIP(ctx, hipEventRecord(ctx->ev[6], st));
    uint64_t nv = 0, nt = 0;
    s = global_marching_cubes<R>(ctx, Q, res, &nv, &nt);
    if (s != SS_OK) return s;
    res->n_vertices = nv;
    res->n_triangles = nt;
    ss_stats& S = res->stats;
    S.ms_total = ev_ms(ctx, 0, 9);
    S.ms_upload = res->host_input ? ev_ms(ctx, 0, 6) : 0.0;
    S.ms_marching_cubes = ev_ms(ctx, 6, 7) + ev_ms(ctx, 8, 9);
    S.ms_stitching = ev_ms(ctx, 7, 8);
    S.n_vertices = nv;
    S.n_triangles = nt;
    res->valid = true;
    res->phase = 2;
    return SS_OK;
}

// neighborhood_search::neighborhood_search_spatial_hashing (neighborhood_search.rs:131-230): lists in the order of the
// reference's sequential function; the result carries only the neighbour lists
template <class R>
ss_status neighborhood_search_impl(ss_context* ctx, const R* xyz, uint64_t n_in, const R domain_min[3], const R domain_max[3], R search_radius, ss_result* res) {
    if (!ctx || !res || !domain_min || !domain_max) return SS_ERR_INVALID_ARGUMENT;
    if (res->ctx != ctx) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context");
    ctx->err.clear();
    ctx->err_detail = 0;
    if (!(search_radius > R(0.0))) return fail(ctx, SS_ERR_UNKNOWN, "search radius for neighborhood search has to be positive (reference: assert)");
    if (n_in >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "too many particles");
    if (n_in && !xyz) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "particle pointer is null");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    ss_status s = ensure_events(ctx);
    if (s != SS_OK) return s;
    hipStream_t st = ctx->stream;
    res->valid = false;
    res->phase = 0;
    reset_host_flags(res);
    memset(&res->stats, 0, sizeof(res->stats));
    res->n_input = res->n_particles = n_in;
    res->n_vertices = res->n_triangles = 0;
    res->has_inside = false;
    res->is_f64 = sizeof(R) == 8;
    res->global_strategy = true;
    memset(&result_grid<R>(res), 0, sizeof(typename TypesOf<R>::grid));
    memset(&result_subgrid<R>(res), 0, sizeof(typename TypesOf<R>::grid));
    typename TypesOf<R>::grid sgrid;
    if (grid_from_aabb<R>(&sgrid, domain_min, domain_max, search_radius) != 0)
        return fail(ctx, SS_ERR_UNKNOWN, "domain for neighborhood search has to be consistent and not degenerate (reference: assert)");
    SSGlobT<R> Q;
    memset(&Q, 0, sizeof(Q));
    double scell_d = 1.0;
    for (int d = 0; d < 3; ++d) {
        Q.smin[d] = sgrid.aabb_min[d];
        if (sgrid.n_cells[d] > 2000000000ll) return fail(ctx, SS_ERR_UNSUPPORTED, "search grid too large");
        Q.snc[d] = (int)sgrid.n_cells[d];
        scell_d *= (double)sgrid.n_cells[d];
    }
    if (scell_d >= 4.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "search grid too large for the dense cell table of this build");
    const R h = search_radius;
    Q.h = h;
    Q.h2 = h * h;
    Q.sigma = R(8.0) / (h * h * h);
    Q.w0 = ss_kernel_evaluate(R(0.0), h, Q.sigma);
    Q.mass = R(1.0);
    Q.n = (uint32_t)n_in;
    SS_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    const R* d_xyz = xyz;
    if (n_in && !is_device_pointer(xyz)) {
        SS_HIP(ctx, ctx->xyz_in.reserve(n_in * 3 * sizeof(R)));
        SS_HIP(ctx, hipMemcpyAsync(ctx->xyz_in.p, xyz, n_in * 3 * sizeof(R), hipMemcpyHostToDevice, st));
        d_xyz = ctx->xyz_in.as<R>();
    }
    s = global_search_and_densities<R>(ctx, Q, d_xyz, res);
    if (s != SS_OK) return s;
    SS_HIP(ctx, hipEventRecord(ctx->ev[4], st));
    SS_HIP(ctx, hipStreamSynchronize(st));
    res->stats.ms_total = ev_ms(ctx, 0, 4);
    res->stats.ms_density = ev_ms(ctx, 3, 4);
    res->stats.n_particles = n_in;
    res->valid = true;
    res->phase = 2;
    return SS_OK;
}

template <class T>
ss_status download(ss_result* r, const DevBuf& d, HostBuf& h, bool& flag, size_t count, const T** out) {
    ss_context* ctx = r->ctx;
    if (!flag) {
        SS_HIP(ctx, hipSetDevice(ctx->device));
        SS_HIP(ctx, h.reserve(count * sizeof(T) + 16));
        if (count) SS_HIP(ctx, hipMemcpyAsync(h.p, d.p, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        flag = true;
    }
    *out = reinterpret_cast<const T*>(h.p);
    return SS_OK;
}

void result_release(ss_result* r) {
    for (DevBuf* b : {&r->rho, &r->posvol, &r->posvol_by_index, &r->perm, &r->inside8, &r->G, &r->blk_minmax, &r->block_slot, &r->active_xyz, &r->mc_xyz, &r->mc_slot, &r->masks,
                      &r->vbase, &r->tbase, &r->vertices, &r->vkeys, &r->tri32, &r->tri64})
        b->release();
    for (HostBuf* b : {&r->h_vertices, &r->h_tri64, &r->h_tri32, &r->h_rho, &r->h_vkeys, &r->h_inside, &r->h_nb_ptr, &r->h_nb_idx}) b->release();
    for (DevBuf* b : {&r->nb_ptr, &r->nb_idx, &r->nb_idx64}) b->release();
}

extern "C" ss_status ss_result_create(ss_context* c, ss_result** out);
extern "C" void ss_result_free(ss_result* r);

template <class R>
ss_status reconstruct_inplace_abi(ss_context* c, const R* xyz, uint64_t n, const typename TypesOf<R>::params* prm, ss_result* inout) {
    if (!c || !inout) return SS_ERR_INVALID_ARGUMENT;
    if (inout->ctx != c) return fail(c, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context");
    c->err.clear();
    c->err_detail = 0;
    return reconstruct_impl<R>(c, xyz, n, prm, inout);
}

template <class R>
ss_status reconstruct_abi(ss_context* c, const R* xyz, uint64_t n, const typename TypesOf<R>::params* prm, ss_result** out) {
    if (!c || !out) return SS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    ss_result* r = nullptr;
    ss_status s = ss_result_create(c, &r);
    if (s != SS_OK) return s;
    s = reconstruct_inplace_abi<R>(c, xyz, n, prm, r);
    if (s != SS_OK) {
        ss_result_free(r);
        return s;
    }
    *out = r;
    return SS_OK;
}

template <class R>
ss_status grid_for_reconstruction_abi(ss_context* c, const R* xyz, uint64_t n_in, const typename TypesOf<R>::params* prm,
                                      typename TypesOf<R>::grid* out) {
    if (!c || !prm || !out) return SS_ERR_INVALID_ARGUMENT;
    c->err.clear();
    if (!(prm->cube_size > R(0.0))) return fail(c, SS_ERR_UNKNOWN, "cube size must be positive");
    if (!(prm->compact_support_radius >= R(0.0))) return fail(c, SS_ERR_UNKNOWN, "compact support radius must be non-negative");
    if (n_in >= (1ull << 31)) return fail(c, SS_ERR_UNSUPPORTED, "too many particles");
    SS_HIP(c, hipSetDevice(c->device));
    R pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
    bool have = false;
    if (!prm->has_particle_aabb && n_in > 0) {
        // note: lib.rs:476-507 computes the AABB over the particles it is given (no filtering here)
        const R* d_xyz = nullptr;
        if (!xyz) return fail(c, SS_ERR_INVALID_ARGUMENT, "particle pointer is null");
        if (is_device_pointer(xyz)) {
            d_xyz = xyz;
        } else {
            SS_HIP(c, c->xyz_in.reserve(n_in * 3 * sizeof(R)));
            SS_HIP(c, hipMemcpyAsync(c->xyz_in.p, xyz, n_in * 3 * sizeof(R), hipMemcpyHostToDevice, c->stream));
            d_xyz = c->xyz_in.as<R>();
        }
        ss_status s = compute_particle_aabb<R>(c, d_xyz, (uint32_t)n_in, pmin, pmax);
        if (s != SS_OK) return s;
        have = true;
    }
    int gerr = grid_for_reconstruction<R>(prm, have, pmin, pmax, out);
    if (gerr) return fail(c, SS_ERR_GRID_CONSTRUCTION, "grid construction failed (uniform_grid.rs:147-169)", gerr);
    return SS_OK;
}

template <class R>
ss_status levelset_box_impl(ss_result* r, const int64_t lo[3], const int64_t extent[3], R* out) {
    ss_context* c = r->ctx;
    for (int d = 0; d < 3; ++d)
        if (extent[d] < 0 || extent[d] > 4096 || lo[d] < -2000000000ll || lo[d] > 2000000000ll) return fail(c, SS_ERR_INVALID_ARGUMENT, "box out of range");
    const size_t tot = (size_t)extent[0] * extent[1] * extent[2];
    if (!tot) return SS_OK;
    SS_HIP(c, hipSetDevice(c->device));
    DevBuf tmp;
    SS_HIP(c, tmp.reserve(tot * sizeof(R)));
    const int l[3] = {(int)lo[0], (int)lo[1], (int)lo[2]}, e[3] = {(int)extent[0], (int)extent[1], (int)extent[2]};
    if (r->global_strategy) {
        // dense array over grid.n_points; small by construction, sliced on the host
        const typename TypesOf<R>::grid& g = result_grid<R>(r);
        const size_t npts = (size_t)g.n_points[0] * g.n_points[1] * g.n_points[2];
        std::vector<R> host(npts ? npts : 1);
        SS_HIP(c, hipMemcpyAsync(host.data(), r->G.p, npts * sizeof(R), hipMemcpyDeviceToHost, c->stream));
        SS_HIP(c, hipStreamSynchronize(c->stream));
        for (int64_t x = 0; x < extent[0]; ++x)
            for (int64_t y = 0; y < extent[1]; ++y)
                for (int64_t z = 0; z < extent[2]; ++z) {
                    const int64_t gx = lo[0] + x, gy = lo[1] + y, gz = lo[2] + z;
                    R v = R(0.0);
                    if (gx >= 0 && gy >= 0 && gz >= 0 && gx < g.n_points[0] && gy < g.n_points[1] && gz < g.n_points[2])
                        v = host[((size_t)gx * g.n_points[1] + gy) * g.n_points[2] + gz];
                    out[((size_t)x * extent[1] + y) * extent[2] + z] = v;
                }
        tmp.release();
        return SS_OK;
    }
    // certified sub-blocks were never evaluated in full or stored (SS_OPTION_FULL_LEVELSET, header): their slots hold no values
    if (r->stats.n_truncated_blocks)
        return fail(c, SS_ERR_INVALID_ARGUMENT,
                    "ss_result_levelset_box: this result holds level-set blocks with certified (never evaluated) sub-blocks; set SS_OPTION_FULL_LEVELSET = 1 "
                    "on the context before the reconstruction to read level-set values");
    if (r->n_active == 0) {
        SS_HIP(c, hipMemsetAsync(tmp.p, 0, tot * sizeof(R), c->stream));
    } else {
        ss_launch_levelset_box<R>(dev_params<R>(r), r->G.as<R>(), r->block_slot.as<uint32_t>(), l, e, tmp.as<R>(), c->stream);
    }
    SS_HIP(c, hipMemcpyAsync(out, tmp.p, tot * sizeof(R), hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    tmp.release();
    return SS_OK;
}

template <class G1, class G2>
void convert_grid(const G1& a, G2* b) {
    for (int d = 0; d < 3; ++d) {
        b->aabb_min[d] = (decltype(b->cell_size))a.aabb_min[d];
        b->aabb_max[d] = (decltype(b->cell_size))a.aabb_max[d];
        b->n_points[d] = a.n_points[d];
        b->n_cells[d] = a.n_cells[d];
    }
    b->cell_size = (decltype(b->cell_size))a.cell_size;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

int ss_abi_version(void) { return SS_ABI_VERSION; }

ss_status ss_context_create(int device_id, ss_context** out) {
    if (!out) return SS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return SS_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= count) return SS_ERR_INVALID_ARGUMENT;
    if (hipSetDevice(device_id) != hipSuccess) return SS_ERR_DEVICE;
    ss_context* c = new (std::nothrow) ss_context();
    if (!c) return SS_ERR_UNKNOWN;
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SS_ERR_DEVICE;
    }
    c->stream = c->own_stream;
    if (const char* e = getenv("SPLASH_K1_OVERLAP")) c->overlap_k1 = e[0] == '1';
    *out = c;
    return SS_OK;
}

void ss_context_destroy(ss_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&c->xyz_in, &c->xyz_filt, &c->flags32, &c->offsets, &c->keys_a, &c->keys_b, &c->vals_a, &c->cell_count,
                      &c->cell_start, &c->pos_sorted, &c->temp, &c->aabb_partial, &c->aabb_out, &c->vcount, &c->tcount, &c->counter, &c->copy_offset, &c->sub_rank,
                      &c->nb_count, &c->nb_tmp, &c->occ_sub, &c->ckeys_a, &c->ckeys_b, &c->cvals_a, &c->cidx, &c->cpos, &c->cell_start2, &c->gboxes, &c->fastdiv_scratch, &c->splat_overflow, &c->mc_nb, &c->splat_tile_idx, &c->splat_tiles, &c->splat_counts, &c->splat_off, &c->splat_bound, &c->splat_trunc, &c->own_flag})
        b->release();
    for (DevBuf& b : c->post_pool) b.release();
    c->zeros.release();
    c->sort_work.release();
    if (c->mail_host) (void)hipHostFree(c->mail_host);
    if (c->ev_ok)
        for (int i = 0; i < 26; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    c->zeros_k1.release();
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* ss_last_error(const ss_context* c) { return c ? c->err.c_str() : "null context"; }
int ss_last_error_detail(const ss_context* c) { return c ? c->err_detail : 0; }

ss_status ss_context_set_option(ss_context* c, int option, int value) {
    if (!c) return SS_ERR_INVALID_ARGUMENT;
    if (option == SS_OPTION_FULL_LEVELSET) {
        c->full_levelset = value != 0;
        return SS_OK;
    }
    if (option == SS_OPTION_WIDEN_ON_DEVICE) {
        c->widen_on_device = value != 0;
        return SS_OK;
    }
    if (option == SS_OPTION_SPLIT_MC_OFFSETS) {
        c->split_mc_offsets = value != 0;
        return SS_OK;
    }
    if (option == SS_OPTION_SPLAT_TWO_PASS) {
        if (value < -1 || value > 1) return fail(c, SS_ERR_INVALID_ARGUMENT, "SS_OPTION_SPLAT_TWO_PASS takes -1 (automatic), 0 or 1");
        c->two_pass = value;
        return SS_OK;
    }
    return fail(c, SS_ERR_INVALID_ARGUMENT, "unknown context option");
}

ss_status ss_measure_hbm_bandwidth(ss_context* c, uint64_t bytes, int repetitions, double* read_gbs, double* copy_gbs) {
    if (!c || !read_gbs || !copy_gbs || repetitions < 1 || bytes < (1ull << 20)) return SS_ERR_INVALID_ARGUMENT;
    SS_HIP(c, hipSetDevice(c->device));
    ss_status s = ensure_events(c);
    if (s != SS_OK) return s;
    hipStream_t st = c->stream;
    const size_t n4 = (size_t)(bytes / 16);
    DevBuf a, b;
    SS_HIP(c, a.reserve(n4 * 16 + 64));
    SS_HIP(c, b.reserve(n4 * 16 + 64));
    SS_HIP(c, hipMemsetAsync(a.p, 0, n4 * 16, st));
    SS_HIP(c, hipMemsetAsync(b.p, 0, n4 * 16, st));
    double best[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; ++mode)
        for (int r = 0; r < repetitions + 1; ++r) {  // (the first launch is a warm-up)
            SS_HIP(c, hipEventRecord(c->ev[0], st));
            ss_launch_stream_probe(mode == 1, a.p, b.p, n4, b.as<float>(), st);
            SS_HIP(c, hipEventRecord(c->ev[1], st));
            SS_HIP(c, hipStreamSynchronize(st));
            const double ms = ev_ms(c, 0, 1);
            const double gbs = (double)(n4 * 16) * (mode == 1 ? 2.0 : 1.0) / (ms * 1.0e6);
            if (r > 0 && gbs > best[mode]) best[mode] = gbs;
        }
    a.release();
    b.release();
    *read_gbs = best[0];
    *copy_gbs = best[1];
    return SS_OK;
}

ss_status ss_context_set_stream(ss_context* c, void* hip_stream) {
    if (!c) return SS_ERR_INVALID_ARGUMENT;
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return SS_OK;
}

ss_status ss_result_create(ss_context* c, ss_result** out) {
    if (!c || !out) return SS_ERR_INVALID_ARGUMENT;
    ss_result* r = new (std::nothrow) ss_result();
    if (!r) return fail(c, SS_ERR_UNKNOWN, "out of host memory");
    r->ctx = c;
    memset(&r->P32, 0, sizeof(r->P32));
    memset(&r->P64, 0, sizeof(r->P64));
    memset(&r->grid32, 0, sizeof(r->grid32));
    memset(&r->sub32, 0, sizeof(r->sub32));
    memset(&r->grid64, 0, sizeof(r->grid64));
    memset(&r->sub64, 0, sizeof(r->sub64));
    memset(&r->stats, 0, sizeof(r->stats));
    *out = r;
    return SS_OK;
}

void ss_result_free(ss_result* r) {
    if (!r) return;
    if (r->ctx) (void)hipSetDevice(r->ctx->device);
    result_release(r);
    delete r;
}

ss_status ss_reconstruct_surface_inplace_f32(ss_context* c, const float* xyz, uint64_t n, const ss_params_f32* prm, ss_result* inout) {
    return reconstruct_inplace_abi<float>(c, xyz, n, prm, inout);
}
ss_status ss_reconstruct_surface_inplace_f64(ss_context* c, const double* xyz, uint64_t n, const ss_params_f64* prm, ss_result* inout) {
    return reconstruct_inplace_abi<double>(c, xyz, n, prm, inout);
}
ss_status ss_reconstruct_surface_f32(ss_context* c, const float* xyz, uint64_t n, const ss_params_f32* prm, ss_result** out) {
    return reconstruct_abi<float>(c, xyz, n, prm, out);
}
ss_status ss_reconstruct_surface_f64(ss_context* c, const double* xyz, uint64_t n, const ss_params_f64* prm, ss_result** out) {
    return reconstruct_abi<double>(c, xyz, n, prm, out);
}

extern "C++" {
namespace {
template <class R>
ss_status shard_begin_abi(ss_context* c, const R* xyz, uint64_t n, const typename TypesOf<R>::params* prm, const typename TypesOf<R>::shard* shard, ss_result* inout) {
    if (!c || !inout || !shard) return SS_ERR_INVALID_ARGUMENT;
    if (inout->ctx != c) return fail(c, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context");
    c->err.clear();
    c->err_detail = 0;
    return phase_begin<R>(c, xyz, n, prm, shard, inout);
}
template <class R>
ss_status shard_get_densities_abi(ss_result* r, R* dst, uint64_t n) {
    if (!r || r->phase < 1 || r->is_f64 != (sizeof(R) == 8) || (!dst && n)) return SS_ERR_INVALID_ARGUMENT;
    ss_context* c = r->ctx;
    if (n != r->n_particles) return fail(c, SS_ERR_INVALID_ARGUMENT, "density count mismatch");
    if (!n) return SS_OK;
    SS_HIP(c, hipSetDevice(c->device));
    SS_HIP(c, hipMemcpyAsync(dst, r->rho.p, n * sizeof(R), hipMemcpyDefault, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}
template <class R>
ss_status shard_set_densities_abi(ss_result* r, const R* src, uint64_t n) {
    if (!r || r->phase != 1 || r->is_f64 != (sizeof(R) == 8) || (!src && n)) return SS_ERR_INVALID_ARGUMENT;
    ss_context* c = r->ctx;
    if (n != r->n_particles) return fail(c, SS_ERR_INVALID_ARGUMENT, "density count mismatch");
    if (!n) return SS_OK;
    SS_HIP(c, hipSetDevice(c->device));
    SS_HIP(c, hipMemcpyAsync(r->rho.p, src, n * sizeof(R), hipMemcpyDefault, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    r->hrho = false;
    return SS_OK;
}
template <class R>
ss_status grid_for_domain_abi(const typename TypesOf<R>::params* prm, const R domain_min[3], const R domain_max[3], typename TypesOf<R>::grid* grid,
                              typename TypesOf<R>::grid* subdomain_grid, R* ghost_margin) {
    if (!prm || !domain_min || !domain_max || !grid || !subdomain_grid) return SS_ERR_INVALID_ARGUMENT;
    if (!(prm->cube_size > R(0.0)) || !(prm->compact_support_radius > R(0.0)) || prm->subdomain_num_cubes_per_dim < 1) return SS_ERR_UNKNOWN;
    typename TypesOf<R>::params p = *prm;
    p.has_particle_aabb = 0;
    typename TypesOf<R>::grid initial;
    if (grid_for_reconstruction<R>(&p, true, domain_min, domain_max, &initial)) return SS_ERR_GRID_CONSTRUCTION;
    R mass = 0, margin = 0;
    initialize_subdomain_parameters<R>(&p, &initial, grid, subdomain_grid, &mass, &margin);
    if (ghost_margin) *ghost_margin = margin;
    return SS_OK;
}
}  // namespace
}  // extern "C++"

ss_status ss_shard_begin_f32(ss_context* c, const float* xyz, uint64_t n, const ss_params_f32* prm, const ss_shard_f32* shard, ss_result* inout) {
    return shard_begin_abi<float>(c, xyz, n, prm, shard, inout);
}
ss_status ss_shard_begin_f64(ss_context* c, const double* xyz, uint64_t n, const ss_params_f64* prm, const ss_shard_f64* shard, ss_result* inout) {
    return shard_begin_abi<double>(c, xyz, n, prm, shard, inout);
}

ss_status ss_shard_finish(ss_context* c, ss_result* inout) {
    if (!c || !inout) return SS_ERR_INVALID_ARGUMENT;
    if (inout->ctx != c) return fail(c, SS_ERR_INVALID_ARGUMENT, "result belongs to a different context");
    c->err.clear();
    return inout->is_f64 ? phase_finish<double>(c, inout) : phase_finish<float>(c, inout);
}

ss_status ss_shard_get_densities(ss_result* r, float* dst, uint64_t n) { return shard_get_densities_abi<float>(r, dst, n); }
ss_status ss_shard_get_densities_f64(ss_result* r, double* dst, uint64_t n) { return shard_get_densities_abi<double>(r, dst, n); }
ss_status ss_shard_set_densities(ss_result* r, const float* src, uint64_t n) { return shard_set_densities_abi<float>(r, src, n); }
ss_status ss_shard_set_densities_f64(ss_result* r, const double* src, uint64_t n) { return shard_set_densities_abi<double>(r, src, n); }

ss_status ss_grid_for_domain_f32(const ss_params_f32* prm, const float domain_min[3], const float domain_max[3], ss_grid_f32* grid,
                                 ss_grid_f32* subdomain_grid, float* ghost_margin) {
    return grid_for_domain_abi<float>(prm, domain_min, domain_max, grid, subdomain_grid, ghost_margin);
}
ss_status ss_grid_for_domain_f64(const ss_params_f64* prm, const double domain_min[3], const double domain_max[3], ss_grid_f64* grid,
                                 ss_grid_f64* subdomain_grid, double* ghost_margin) {
    return grid_for_domain_abi<double>(prm, domain_min, domain_max, grid, subdomain_grid, ghost_margin);
}

ss_status ss_marching_cubes_f32(ss_context* c, const float* values, const int64_t n_points[3], float iso_surface_threshold, float cube_size, const float translation[3],
                                ss_result* inout) {
    return marching_cubes_impl<float>(c, values, n_points, iso_surface_threshold, cube_size, translation, inout);
}
ss_status ss_marching_cubes_f64(ss_context* c, const double* values, const int64_t n_points[3], double iso_surface_threshold, double cube_size,
                                const double translation[3], ss_result* inout) {
    return marching_cubes_impl<double>(c, values, n_points, iso_surface_threshold, cube_size, translation, inout);
}
ss_status ss_neighborhood_search_f32(ss_context* c, const float* xyz, uint64_t n, const float domain_min[3], const float domain_max[3], float search_radius,
                                     ss_result* inout) {
    return neighborhood_search_impl<float>(c, xyz, n, domain_min, domain_max, search_radius, inout);
}
ss_status ss_neighborhood_search_f64(ss_context* c, const double* xyz, uint64_t n, const double domain_min[3], const double domain_max[3], double search_radius,
                                     ss_result* inout) {
    return neighborhood_search_impl<double>(c, xyz, n, domain_min, domain_max, search_radius, inout);
}

ss_status ss_grid_for_reconstruction_f32(ss_context* c, const float* xyz, uint64_t n_in, const ss_params_f32* prm, ss_grid_f32* out) {
    return grid_for_reconstruction_abi<float>(c, xyz, n_in, prm, out);
}
ss_status ss_grid_for_reconstruction_f64(ss_context* c, const double* xyz, uint64_t n_in, const ss_params_f64* prm, ss_grid_f64* out) {
    return grid_for_reconstruction_abi<double>(c, xyz, n_in, prm, out);
}

int ss_result_is_f64(const ss_result* r) { return (r && r->is_f64) ? 1 : 0; }

ss_status ss_result_counts(const ss_result* r, uint64_t* nv, uint64_t* nt) {
    if (!r || !r->valid) return SS_ERR_INVALID_ARGUMENT;
    if (nv) *nv = r->n_vertices;
    if (nt) *nt = r->n_triangles;
    return SS_OK;
}

ss_status ss_result_vertices_f64(ss_result* r, const double** xyz, uint64_t* n) {
    if (!r || !r->valid || !r->is_f64 || !xyz || !n) return SS_ERR_INVALID_ARGUMENT;
    *n = r->n_vertices;
    return download<double>(r, r->vertices, r->h_vertices, r->hv, (size_t)r->n_vertices * 3, xyz);
}

ss_status ss_result_vertices(ss_result* r, const float** xyz, uint64_t* n) {
    if (!r || !r->valid || r->is_f64 || !xyz || !n) return SS_ERR_INVALID_ARGUMENT;
    *n = r->n_vertices;
    return download<float>(r, r->vertices, r->h_vertices, r->hv, (size_t)r->n_vertices * 3, xyz);
}

ss_status ss_result_triangles_u32(ss_result* r, const uint32_t** idx, uint64_t* m) {
    if (!r || !r->valid || !idx || !m) return SS_ERR_INVALID_ARGUMENT;
    *m = r->n_triangles;
    return download<uint32_t>(r, r->tri32, r->h_tri32, r->ht32, (size_t)r->n_triangles * 3, idx);
}

ss_status ss_result_triangles(ss_result* r, const uint64_t** idx, uint64_t* m) {
    if (!r || !r->valid || !idx || !m) return SS_ERR_INVALID_ARGUMENT;
    ss_context* c = r->ctx;
    *m = r->n_triangles;
    const size_t cnt = (size_t)r->n_triangles * 3;
    if (!r->ht64 && cnt >= ((size_t)1 << 20) && !c->widen_on_device) {
        // Large meshes: the indices cross PCIe as u32 (half the bytes of [usize; 3]) in chunks, and host threads widen each chunk into
        // the pinned u64 buffer while the next one is on its way -- the link, not the widening, sets the time.
        SS_HIP(c, hipSetDevice(c->device));
        SS_HIP(c, r->h_tri32.reserve(cnt * 4 + 16));
        SS_HIP(c, r->h_tri64.reserve(cnt * 8 + 16));
#ifndef SS_WIDEN_THREADS
#define SS_WIDEN_THREADS 16
#endif
#ifndef SS_WIDEN_CHUNKS
#define SS_WIDEN_CHUNKS 8
#endif
        constexpr int n_chunks = SS_WIDEN_CHUNKS;
        hipEvent_t ev[n_chunks];
        size_t off[n_chunks + 1];
        for (int k = 0; k <= n_chunks; ++k) off[k] = std::min(cnt, (cnt * (size_t)k / n_chunks + 1023) / 1024 * 1024);  // chunk borders on multiples of 1024 elements
        off[n_chunks] = cnt;
        uint32_t* h32 = reinterpret_cast<uint32_t*>(r->h_tri32.p);
        unsigned long long* h64 = reinterpret_cast<unsigned long long*>(r->h_tri64.p);
        int n_ev = 0;
        hipError_t err = hipSuccess;
        for (int k = 0; k < n_chunks && err == hipSuccess; ++k) {
            err = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            if (err != hipSuccess) break;
            ++n_ev;
            if (off[k + 1] > off[k] && !r->ht32)  // (skipped when ss_result_triangles_u32 has brought the indices to the host already)
                err = hipMemcpyAsync(h32 + off[k], r->tri32.as<uint32_t>() + off[k], (off[k + 1] - off[k]) * 4, hipMemcpyDeviceToHost, c->stream);
            if (err == hipSuccess) err = hipEventRecord(ev[k], c->stream);
        }
        bool threads_ok = true;
        if (err == hipSuccess) {
          try {  // (std::thread / std::vector may throw: nothing may leave an extern "C" function)
            const unsigned hw = std::thread::hardware_concurrency();
            const int n_threads = (int)std::max(1u, std::min((unsigned)SS_WIDEN_THREADS, hw ? hw / 4u : 4u));
            const int device = c->device;
            std::vector<std::thread> pool;
            std::vector<int> failed((size_t)n_threads, 0);
            struct Joiner {  // joins whatever was started, also when a later emplace_back throws
                std::vector<std::thread>& p;
                ~Joiner() {
                    for (auto& th : p)
                        if (th.joinable()) th.join();
                }
            } joiner{pool};
            pool.reserve((size_t)n_threads);
            for (int t = 0; t < n_threads; ++t)
                pool.emplace_back([=, &failed]() {
                    if (hipSetDevice(device) != hipSuccess) {
                        failed[(size_t)t] = 1;
                        return;
                    }
                    for (int k = 0; k < n_chunks; ++k) {
                        if (hipEventSynchronize(ev[k]) != hipSuccess) {
                            failed[(size_t)t] = 1;
                            return;
                        }
                        const size_t len = off[k + 1] - off[k];
                        const size_t b = off[k] + len * (size_t)t / (size_t)n_threads, e = off[k] + len * (size_t)(t + 1) / (size_t)n_threads;
                        const uint32_t* __restrict__ src = h32;
                        unsigned long long* __restrict__ dst = h64;
                        // non-temporal stores: the u64 buffer is written once and read by the caller later -- a cached store would first READ every
                        // line it overwrites (345 MB more traffic on a 14 M-triangle mesh) and evict the u32 chunk that is being read
                        for (size_t i = b; i < e; ++i) __builtin_nontemporal_store((unsigned long long)src[i], dst + i);
                    }
                });
            for (auto& th : pool) th.join();
            for (int f : failed)
                if (f) err = hipErrorUnknown;
          } catch (...) {
            threads_ok = false;  // no threads to be had: the device widens instead (below)
          }
        }
        if (err == hipSuccess) err = hipStreamSynchronize(c->stream);
        for (int k = 0; k < n_ev; ++k) (void)hipEventDestroy(ev[k]);
        if (err != hipSuccess) return fail(c, SS_ERR_DEVICE, std::string("triangle download failed: ") + hipGetErrorString(err));
        if (threads_ok) {
            r->ht64 = true;
            r->ht32 = true;  // (the u32 indices are on the host as well now)
        }
    }
    if (!r->ht64) {
        // small meshes: widen on the device, one copy
        SS_HIP(c, hipSetDevice(c->device));
        SS_HIP(c, r->tri64.reserve(cnt * 8 + 16));
        ss_launch_widen(r->tri32.as<uint32_t>(), cnt, r->tri64.as<unsigned long long>(), c->stream);
    }
    const unsigned long long* p = nullptr;
    ss_status s = download<unsigned long long>(r, r->tri64, r->h_tri64, r->ht64, (size_t)r->n_triangles * 3, &p);
    *idx = reinterpret_cast<const uint64_t*>(p);
    return s;
}

ss_status ss_result_vertex_keys(ss_result* r, const uint64_t** keys, uint64_t* n) {
    if (!r || !r->valid || !keys || !n) return SS_ERR_INVALID_ARGUMENT;
    *n = r->n_vertices;
    const unsigned long long* p = nullptr;
    ss_status s = download<unsigned long long>(r, r->vkeys, r->h_vkeys, r->hkeys, (size_t)r->n_vertices, &p);
    *keys = reinterpret_cast<const uint64_t*>(p);
    return s;
}

ss_status ss_result_grid(const ss_result* r, ss_grid_f32* out) {
    if (!r || !r->valid || !out) return SS_ERR_INVALID_ARGUMENT;
    if (r->is_f64)
        convert_grid(r->grid64, out);  // rounded to f32
    else
        *out = r->grid32;
    return SS_OK;
}

ss_status ss_result_grid_f64(const ss_result* r, ss_grid_f64* out) {
    if (!r || !r->valid || !out) return SS_ERR_INVALID_ARGUMENT;
    if (r->is_f64)
        *out = r->grid64;
    else
        convert_grid(r->grid32, out);  // exact
    return SS_OK;
}

ss_status ss_result_subdomain_grid_f64(const ss_result* r, ss_grid_f64* out, int32_t* present) {
    if (!r || !r->valid || !out || !present) return SS_ERR_INVALID_ARGUMENT;
    if (r->is_f64)
        *out = r->sub64;
    else
        convert_grid(r->sub32, out);
    *present = r->global_strategy ? 0 : 1;  // None for the global strategy (lib.rs:249-250)
    return SS_OK;
}

ss_status ss_result_subdomain_grid(const ss_result* r, ss_grid_f32* out, int32_t* present) {
    if (!r || !r->valid || !out || !present) return SS_ERR_INVALID_ARGUMENT;
    if (r->is_f64)
        convert_grid(r->sub64, out);
    else
        *out = r->sub32;
    *present = r->global_strategy ? 0 : 1;
    return SS_OK;
}

ss_status ss_result_particle_densities_f64(ss_result* r, const double** rho, uint64_t* n) {
    if (!r || !r->valid || !r->is_f64 || !rho || !n) return SS_ERR_INVALID_ARGUMENT;
    *n = r->n_particles;
    return download<double>(r, r->rho, r->h_rho, r->hrho, (size_t)r->n_particles, rho);
}

ss_status ss_result_particle_densities(ss_result* r, const float** rho, uint64_t* n) {
    if (!r || !r->valid || r->is_f64 || !rho || !n) return SS_ERR_INVALID_ARGUMENT;
    *n = r->n_particles;
    return download<float>(r, r->rho, r->h_rho, r->hrho, (size_t)r->n_particles, rho);
}

ss_status ss_result_particle_inside_aabb(ss_result* r, const uint8_t** flags, uint64_t* n) {
    if (!r || !r->valid || !flags || !n) return SS_ERR_INVALID_ARGUMENT;
    if (!r->has_inside) {
        *flags = nullptr;
        *n = 0;
        return SS_OK;
    }
    *n = r->n_input;
    if (r->n_input == 0) {
        static const uint8_t dummy = 0;
        *flags = &dummy;
        return SS_OK;
    }
    return download<uint8_t>(r, r->inside8, r->h_inside, r->hinside, (size_t)r->n_input, flags);
}

ss_status ss_result_particle_neighbors(ss_result* r, const uint64_t** row_ptr, const uint64_t** neighbors, uint64_t* n_particles) {
    if (!r || !r->valid || !row_ptr || !neighbors || !n_particles) return SS_ERR_INVALID_ARGUMENT;
    ss_context* c = r->ctx;
    *n_particles = r->n_particles;
    if (!r->has_neighbors) {  // Option::None (global_neighborhood_list was not requested)
        *row_ptr = nullptr;
        *neighbors = nullptr;
        return SS_OK;
    }
    if (!r->hnbi && r->n_neighbors) {
        SS_HIP(c, hipSetDevice(c->device));
        SS_HIP(c, r->nb_idx64.reserve((size_t)r->n_neighbors * 8 + 16));
        ss_launch_widen(r->nb_idx.as<uint32_t>(), (size_t)r->n_neighbors, r->nb_idx64.as<unsigned long long>(), c->stream);
    }
    const unsigned long long *p = nullptr, *q = nullptr;
    ss_status s = download<unsigned long long>(r, r->nb_ptr, r->h_nb_ptr, r->hnbp, (size_t)r->n_particles + 1, &p);
    if (s != SS_OK) return s;
    s = download<unsigned long long>(r, r->nb_idx64, r->h_nb_idx, r->hnbi, (size_t)r->n_neighbors, &q);
    if (s != SS_OK) return s;
    *row_ptr = reinterpret_cast<const uint64_t*>(p);
    *neighbors = reinterpret_cast<const uint64_t*>(q);
    return SS_OK;
}

ss_status ss_result_stats(const ss_result* r, ss_stats* out) {
    if (!r || !r->valid || !out) return SS_ERR_INVALID_ARGUMENT;
    *out = r->stats;
    return SS_OK;
}

ss_status ss_result_device_vertices(const ss_result* r, const float** d, uint64_t* n) {
    if (!r || !r->valid || r->is_f64 || !d || !n) return SS_ERR_INVALID_ARGUMENT;
    *d = r->vertices.as<float>();
    *n = r->n_vertices;
    return SS_OK;
}
ss_status ss_result_device_triangles_u32(const ss_result* r, const uint32_t** d, uint64_t* n) {
    if (!r || !r->valid || !d || !n) return SS_ERR_INVALID_ARGUMENT;
    *d = r->tri32.as<uint32_t>();
    *n = r->n_triangles;
    return SS_OK;
}
ss_status ss_result_device_particle_densities(const ss_result* r, const float** d, uint64_t* n) {
    if (!r || !r->valid || r->is_f64 || !d || !n) return SS_ERR_INVALID_ARGUMENT;
    *d = r->rho.as<float>();
    *n = r->n_particles;
    return SS_OK;
}

ss_status ss_result_levelset_box(ss_result* r, const int64_t lo[3], const int64_t extent[3], float* out) {
    if (!r || !r->valid || r->is_f64 || !lo || !extent || !out) return SS_ERR_INVALID_ARGUMENT;
    return levelset_box_impl<float>(r, lo, extent, out);
}

ss_status ss_result_levelset_box_f64(ss_result* r, const int64_t lo[3], const int64_t extent[3], double* out) {
    if (!r || !r->valid || !r->is_f64 || !lo || !extent || !out) return SS_ERR_INVALID_ARGUMENT;
    return levelset_box_impl<double>(r, lo, extent, out);
}

ss_status ss_result_debug_certified(ss_result* r, uint32_t* masks, uint32_t* block_xyz, uint64_t capacity, uint64_t* n_active) {
    if (!r || !r->valid || !n_active || r->global_strategy) return SS_ERR_INVALID_ARGUMENT;
    ss_context* ctx = r->ctx;
    *n_active = r->n_active;
    const uint64_t n = std::min<uint64_t>(capacity, r->n_active);
    if (!n) return SS_OK;
    if (!masks || !block_xyz) return SS_ERR_INVALID_ARGUMENT;
    // the masks live in the context's scratch: another reconstruction on the context (with any result object) has overwritten or freed them
    if (r->dbg_certified && r->dbg_serial != ctx->call_serial)
        return fail(ctx, SS_ERR_INVALID_ARGUMENT, "ss_result_debug_certified: the certificates of this result are gone (another call ran on its context since)");
    SS_HIP(ctx, hipSetDevice(ctx->device));
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (r->dbg_certified)
        SS_HIP(ctx, hipMemcpy(masks, r->dbg_certified, n * 4, hipMemcpyDeviceToHost));
    else
        memset(masks, 0, n * 4);  // (every block was evaluated completely)
    SS_HIP(ctx, hipMemcpy(block_xyz, r->active_xyz.p, n * 12, hipMemcpyDeviceToHost));
    return SS_OK;
}

ss_status ss_result_subdomain_stats(ss_result* r, uint64_t* n_occupied, uint64_t* n_sub_particles) {
    if (!r || !r->valid || !n_occupied || !n_sub_particles) return SS_ERR_INVALID_ARGUMENT;
    *n_occupied = r->n_occupied_subdomains;
    *n_sub_particles = r->n_subdomain_particles;
    return SS_OK;
}

}  // extern "C"
