// ss_device.h -- shared host/device definitions for the gfx950 surface-reconstruction kernels.
//
// All floating point code in this library is compiled with -ffp-contract=off and HIP's default
// correctly-rounded divide/sqrt, so that every expression below evaluates exactly like the
// reference's scalar Rust code (IEEE, no FMA contraction, SURVEY.md section 8 preamble).
// Everything is templated on the reference's `Real` type R (float: reconstruct_surface::<i64,f32>,
// double: ::<i64,f64>); literals are written R(x) like the reference's `R::from_float(x)`.
// Citations are relative to /root/reference/splashsurf_lib/src/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <limits>

#define SS_BLOCK 8              // level-set block edge in grid points (8^3 = 512 points)
#define SS_BLOCK_POINTS 512
// Position of point (x, y, z), 0 <= x, y, z < 8, inside a level-set block: the eight 4x4x4 sub-blocks one after the other (sub-block
// (sx, sy, sz) -> index (sx << 2) | (sy << 1) | sz), the 64 points of a sub-block in (x, y, z) order -- a wave stores its sub-block
// as one 256-byte run.
#define SS_BLOCK_OFFSET(x, y, z) ((((((x) >> 2) << 2) | (((y) >> 2) << 1) | ((z) >> 2)) << 6) | (((x) & 3) << 4) | (((y) & 3) << 2) | ((z) & 3))
#define SS_MAX_ROWS 256         // (x,y) search-cell rows per batch while gathering a tile
#define SS_WTILE 384            // tile entries one wave orders in LDS (k_splat_gather) = chunk of the accumulate kernel

// Arithmetic of the level-set accumulation G += V * W(|x - p|)  (Parameters::enable_simd, lib.rs:179-181):
//   GENERIC / FAST   the reference's scalar loop (dense_subdomains.rs:784-847, kernel.rs:58-107), bit for bit; FAST = lean exact
//                    sqrt + device-verified reciprocal division, GENERIC = hipcc's correctly rounded sqrt / divide
//   SIMD*            the reference's AVX2+FMA loop (dense_subdomains.rs:991-1133, kernel.rs:319-378) applied to every (particle,
//                    point) pair: d^2 by two fma, mask d^2 < h^2, q = r * (1/h), v = max(1 - q, 0), fma polynomial, fma accumulate.
//                    SIMD / SIMD_LEAN use a correctly rounded sqrt like _mm256_sqrt_ps (generic / lean variant), SIMD_HW the raw
//                    v_sqrt_f32 (<= 1 ulp), requested with enable_simd = 2.
enum : int { SS_ARITH_GENERIC = 0, SS_ARITH_FAST = 1, SS_ARITH_SIMD = 2, SS_ARITH_SIMD_LEAN = 3, SS_ARITH_SIMD_HW = 4,
              SS_ARITH_BOUND = 5 /* classification pass only: SIMD_HW without the reach test (the clamp makes far terms 0) */ };

// out = c - x with the result clamped to [0, 1] by the subtraction's output modifier (ONE VALU instruction, c an inline constant), and a
// comment in the instruction stream that keeps hipcc from if-converting a wave-uniform branch.  SS_HIP_EMU is defined by tests/emu only: the
// host build that runs these kernels on the CPU for the no-GPU tests (never by the product's build) spells the instruction out in C.
#ifdef SS_HIP_EMU
#define SS_SUB_CLAMP(out, c, x) (out) = emu_clamp01((float)(c) - (float)(x))
#define SS_ASM_NOTE(text)
#else
#define SS_SUB_CLAMP(out, c, x) asm("v_sub_f32_e64 %0, " #c ", %1 clamp" : "=v"(out) : "v"(x))
#define SS_ASM_NOTE(text) asm volatile("; " text ::)
#endif

// candidate particles (4-byte index keys) held in LDS per pass of the large-tile splat kernel
template <class R> struct SSTileCap { static constexpr int value = 8192; };
// tile entries one wave of the wave-per-block splat kernel holds in LDS (k_splat_fused); blocks with more candidates take the arena path
template <class R>
struct SSWaveChunk {
    static constexpr int value = sizeof(R) == 4 ? 192 : 128;  // whole 64-entry batches
};
// Tiles of SS_WTILE < n <= SS_SORT_TILE_MAX entries are written by k_splat_gather_large in scan order with their particle indices;
// k_splat_accumulate_list orders them in LDS, and only for the blocks that need an exact sum.
#define SS_SORT_TILE_MAX 4096

template <class R> struct SSVec;
template <> struct SSVec<float> { using v2 = float2; using v4 = float4; };
template <> struct SSVec<double> { using v2 = double2; using v4 = double4; };
template <class R> using ss_real2 = typename SSVec<R>::v2;
template <class R> using ss_real4 = typename SSVec<R>::v4;

// positions in sorted order (the K1 cell order, the per-subdomain copies of the density stage): three values per particle, 12 / 24 bytes --
// the density kernel is bound by the vector L1, and the fourth lane of a float4 was padding.  -DSS_POS4 restores the padded form (A/B).
template <class R> struct ss_real3 { R x, y, z; };
#ifdef SS_POS4
template <class R> using ss_pos = typename SSVec<R>::v4;
#else
template <class R> using ss_pos = ss_real3<R>;
#endif
__host__ __device__ inline float4 ss_make4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__host__ __device__ inline double4 ss_make4(double x, double y, double z, double w) { return make_double4(x, y, z, w); }
__host__ __device__ inline float2 ss_make2(float x, float y) { return make_float2(x, y); }
__host__ __device__ inline double2 ss_make2(double x, double y) { return make_double2(x, y); }

template <class R>
__host__ __device__ inline ss_pos<R> ss_make_pos(R x, R y, R z) {
#ifdef SS_POS4
    return ss_make4(x, y, z, R(0.0));
#else
    return ss_pos<R>{x, y, z};
#endif
}

__host__ __device__ inline float ss_floor(float x) { return floorf(x); }
__host__ __device__ inline double ss_floor(double x) { return floor(x); }
__host__ __device__ inline float ss_ceil(float x) { return ceilf(x); }
__host__ __device__ inline double ss_ceil(double x) { return ceil(x); }
__host__ __device__ inline float ss_sqrt(float x) { return sqrtf(x); }
__host__ __device__ inline double ss_sqrt(double x) { return sqrt(x); }
__host__ __device__ inline float ss_min(float a, float b) { return fminf(a, b); }
__host__ __device__ inline double ss_min(double a, double b) { return fmin(a, b); }
__host__ __device__ inline float ss_max(float a, float b) { return fmaxf(a, b); }
__host__ __device__ inline double ss_max(double a, double b) { return fmax(a, b); }

template <class R>
struct SSDevT {
    // global (padded) marching-cubes grid: dense_subdomains.rs:183-188
    R gmin[3];
    R cs;
    int np[3];  // points per dim
    int nc[3];  // cells per dim
    // subdomain grid (same min): dense_subdomains.rs:207-213
    int n_sub_cubes;
    int ns[3];
    R sub_size;
    int sub_radius;  // ceil(margin / sub_size): subdomains to check in each direction (dense_subdomains.rs:1827-1832)
    int sc[3];       // per-subdomain neighbourhood-search grid: table dims (cells of edge h, upper bound)
    // SPH kernel and level-set constants
    R h;          // compact support radius
    double inv_h; // 1/h in double: block -> search-cell ranges without a division per wave
    R h2;         // h*h                       (neighborhood_search.rs:367)
    R H2;         // h*h*1.01                  (dense_subdomains.rs:1224-1226)
    R sigma;      // 8/(h*h*h)                 (kernel.rs:60-68)
    R w0;         // W(0)
    R mass;       // rest mass                 (dense_subdomains.rs:117-118)
    R threshold;  // iso-surface threshold
    R margin;     // ghost particle margin     (dense_subdomains.rs:120-121)
    R reach;      // conservative reach of a particle: sqrt(support^2)*(1+1e-4), support^2 = 1.01 h^2 (scalar) or h^2 (SIMD arithmetic)
    R R2;         // support^2 * (1+1e-4): squared reach of the conservative block / wave filters of the splat
    R R2near;     // (0.64 h)^2: sub-block filter of the splat's classification pass (make_device_params)
    R thr_inside; // threshold * (1 + 1e-4): a lower bound above it certifies 'inside' whatever the summation order
    // certificate on the matrix pipe (splat_cert_record): C4 sigma (1 - 2e-5), and the slack eps = cert_e1 (|px| + |py| + |pz|) + cert_e0 taken off 1 - |p|^2
    R cert_vscale, cert_e1, cert_e0;
    R cert_e1s, cert_e0s;  // the same slack for records relative to a SUB-BLOCK's centre (largest point coordinate 1.5 cs / h; k_splat_certify_big)
    // Parameters::enable_simd: constants of CubicSplineKernelAvxF32 (kernel.rs:327-337), formed in f32 like the reference does
    R avx_inv_h, avx_sigma, avx_sigma2, avx_sigma6, avx_sigma12;
    int arith;    // SS_ARITH_* of the level-set accumulation chosen for this call
    R coord_slack;  // absolute slack covering rounding of coordinates in conservative tests
    // Dense array of SPLAT CELLS, absolute cell coordinate K in [kmin, kmin+kdim): the cells the particles are sorted by for the level-set
    // splat (k_cell_keys; the densities use per-subdomain copies in the reference's own search grids).  They are aligned with the lattice of
    // level-set blocks: edge 8 cs / sk, cell 0 starts at sorg, and the box of block b's points dilated by the particle reach is covered on
    // every axis by exactly the cells [sk b, sk b + sn1) -- the (x, y) rows of cells a block scans and their z extents are the same for
    // every block (splat_row_cells), no per-block geometry.  sk is chosen so that a cell holds about one particle at SPH rest spacing.
    int kmin[3];
    int kdim[3];
    double sorg[3];  // lower corner of splat cell 0 per axis
    double sinv;     // 1 / cell edge
    int sk;          // cells per block pitch (8 grid points)
    int sn1;         // cells per axis that cover a block's dilated box
    float so, se, srho;  // in units of cs relative to a block's first point: lower face of its first covering cell, cell edge, padded reach
    // level-set blocks: nb = blocks of the whole grid; the dense per-block tables (slots, flags, MC slots) cover the window
    // [bt_org, bt_org + bt_dim) only -- the whole grid for a single-process job, the shard's blocks (+ one layer above, which
    // marching cubes looks into) for a rank of a multi-GPU job: the tables and every pass over them scale with the brick, not the domain
    int nb[3];
    int bt_org[3], bt_dim[3];
    // shard (multi-GPU): this process reconstructs the subdomains [sub_lo, sub_hi) only.  Full domain: [0, ns).
    int sub_lo[3], sub_hi[3];
    int pt_lo[3], pt_hi[3];      // grid points of the shard region, inclusive: [sub_lo*n, min(np-1, sub_hi*n)]
    int blk_lo[3], blk_hi[3];    // level-set blocks covering those points, inclusive
    uint32_t n;  // particle count (after the AABB filter)
};

// ---- cubic spline kernel, scalar path of the reference (kernel.rs:71-81, 103-106) ----
template <class R>
__host__ __device__ inline R ss_cubic_function(R q) {
    const R pi = R(3.14159265358979323846);
    if (q < R(1.0)) {
        return (R(3.0) / (R(2.0) * pi)) * ((R(2.0) / R(3.0)) - q * q + R(0.5) * q * q * q);
    } else if (q < R(2.0)) {
        R x = R(2.0) - q;
        return (R(1.0) / (R(4.0) * pi)) * x * x * x;
    }
    return R(0.0);
}

template <class R>
__host__ __device__ inline R ss_kernel_evaluate(R r, R h, R sigma) {
    R q = (r + r) / h;
    return sigma * ss_cubic_function<R>(q);
}

// ---- subdomain that CONTAINS a point in the half-open sense (aabb.rs:220-222 applied to
//      subdomain_grid.cell_aabb, uniform_grid.rs:454-467); used to file particles into the global
//      search cells.  Corner coordinates of adjacent subdomains are bit-identical, so exactly one
//      subdomain contains a point per axis. ----
template <class R>
__host__ __device__ inline int ss_container_subdomain_axis(const SSDevT<R>& P, R x, int d) {
    R normalized = (x - P.gmin[d]) / P.sub_size;
    int s = (int)ss_floor(normalized);
    R lo = P.gmin[d] + (R)s * P.sub_size;
    R hi = P.gmin[d] + (R)(s + 1) * P.sub_size;
    if (x < lo)
        s -= 1;
    else if (!(x < hi))
        s += 1;
    return s;
}

// ---- absolute search-cell coordinate of x in the neighbourhood-search grid the reference builds for
//      subdomain s (neighborhood_search.rs:370 on the AABB of dense_subdomains.rs:560-565;
//      uniform_grid.rs:189-201 alignment; uniform_grid.rs:444-451 enclosing_cell) ----
template <class R>
__host__ __device__ inline int ss_search_cell_axis(const SSDevT<R>& P, int s, R x, int d) {
    R amin = P.gmin[d] + (R)s * P.sub_size;
    R mmin = amin - P.margin * R(1.5);
    R kf = ss_floor(mmin / P.h);
    R aligned = kf * P.h;
    R c = ss_floor((x - aligned) / P.h);
    return (int)kf + (int)c;
}

// splat cell of a coordinate (per axis), clamped into the dense cell array.  (Clamping can only trigger for a rank of a multi-GPU job,
// whose table covers its brick's blocks: a held particle beyond it is in reach of none of them, and every candidate of a block is
// tested individually anyway.)
template <class R>
__host__ __device__ inline int ss_splat_cell_axis(const SSDevT<R>& P, R x, int d) {
    const double c = floor(((double)x - P.sorg[d]) * P.sinv);
    const double lo = (double)P.kmin[d], hi = (double)(P.kmin[d] + P.kdim[d] - 1);
    return (int)(c < lo ? lo : (c > hi ? hi : c));
}

template <class R>
__host__ __device__ inline uint32_t ss_cell_key(const SSDevT<R>& P, int kx, int ky, int kz) {
    return (uint32_t)(((int64_t)(kx - P.kmin[0]) * P.kdim[1] + (ky - P.kmin[1])) * P.kdim[2] + (kz - P.kmin[2]));
}

// position of level-set block (bx, by, bz) in the dense per-block tables; blocks outside the window have no entry (and no values)
template <class R>
__host__ __device__ inline bool ss_block_in_table(const SSDevT<R>& P, int bx, int by, int bz) {
    return bx >= P.bt_org[0] && by >= P.bt_org[1] && bz >= P.bt_org[2] && bx < P.bt_org[0] + P.bt_dim[0] && by < P.bt_org[1] + P.bt_dim[1] &&
           bz < P.bt_org[2] + P.bt_dim[2];
}
template <class R>
__host__ __device__ inline size_t ss_block_index(const SSDevT<R>& P, int bx, int by, int bz) {
    return ((size_t)(bx - P.bt_org[0]) * (size_t)P.bt_dim[1] + (size_t)(by - P.bt_org[1])) * (size_t)P.bt_dim[2] + (size_t)(bz - P.bt_org[2]);
}
template <class R>
__host__ __device__ inline void ss_block_of_index(const SSDevT<R>& P, uint32_t b, int* bx, int* by, int* bz) {
    *bz = P.bt_org[2] + (int)(b % (uint32_t)P.bt_dim[2]);
    *by = P.bt_org[1] + (int)((b / (uint32_t)P.bt_dim[2]) % (uint32_t)P.bt_dim[1]);
    *bx = P.bt_org[0] + (int)(b / ((uint32_t)P.bt_dim[2] * (uint32_t)P.bt_dim[1]));
}

template <class R>
__host__ __device__ inline bool ss_cell_in_range(const SSDevT<R>& P, int kx, int ky, int kz) {
    return kx >= P.kmin[0] && ky >= P.kmin[1] && kz >= P.kmin[2] && kx < P.kmin[0] + P.kdim[0] &&
           ky < P.kmin[1] + P.kdim[1] && kz < P.kmin[2] + P.kdim[2];
}
