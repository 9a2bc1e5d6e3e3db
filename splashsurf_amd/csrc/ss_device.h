// ss_device.h -- shared host/device definitions for the gfx950 surface-reconstruction kernels.
//
// All floating point code in this library is compiled with -ffp-contract=off and HIP's default
// correctly-rounded f32 divide/sqrt, so that every expression below evaluates exactly like the
// reference's scalar Rust code (IEEE f32, no FMA contraction, SURVEY.md section 8 preamble).
// Citations are relative to /root/reference/splashsurf_lib/src/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SS_BLOCK 8              // level-set block edge in grid points (8^3 = 512 points, 2 KiB of f32)
#define SS_BLOCK_POINTS 512
#define SS_TILE_CAP 2048        // candidate particles staged in LDS per pass of the splat kernel
#define SS_MAX_ROWS 256         // (x,y) search-cell rows per batch while gathering a tile

struct SSDev {
    // global (padded) marching-cubes grid: dense_subdomains.rs:183-188
    float gmin[3];
    float cs;
    int np[3];  // points per dim
    int nc[3];  // cells per dim
    // subdomain grid (same min): dense_subdomains.rs:207-213
    int n_sub_cubes;
    int ns[3];
    float sub_size;
    int sub_radius;  // ceil(margin / sub_size): subdomains to check in each direction (dense_subdomains.rs:1827-1832)
    int sc[3];       // per-subdomain neighbourhood-search grid: table dims (cells of edge h, upper bound)
    // SPH kernel and level-set constants
    float h;          // compact support radius
    float h2;         // h*h                       (neighborhood_search.rs:367)
    float H2;         // h*h*1.01                  (dense_subdomains.rs:1224-1226)
    float sigma;      // 8/(h*h*h)                 (kernel.rs:60-68)
    float w0;         // W(0)
    float mass;       // rest mass                 (dense_subdomains.rs:117-118)
    float threshold;  // iso-surface threshold
    float margin;     // ghost particle margin     (dense_subdomains.rs:120-121)
    float reach;      // conservative reach of a particle: sqrt(1.01)*h*(1+1e-4)
    float coord_slack;  // absolute slack covering f32 rounding of coordinates in conservative tests
    // dense array of search cells (edge h), absolute cell coordinate K in [kmin, kmin+kdim)
    int kmin[3];
    int kdim[3];
    // level-set blocks
    int nb[3];
    // shard (multi-GPU): this process reconstructs the subdomains [sub_lo, sub_hi) only.  Full domain: [0, ns).
    int sub_lo[3], sub_hi[3];
    int pt_lo[3], pt_hi[3];      // grid points of the shard region, inclusive: [sub_lo*n, min(np-1, sub_hi*n)]
    int blk_lo[3], blk_hi[3];    // level-set blocks covering those points, inclusive
    uint32_t n;  // particle count (after the AABB filter)
};

// ---- cubic spline kernel, scalar path of the reference (kernel.rs:71-81, 103-106) ----
__host__ __device__ inline float ss_cubic_function(float q) {
    const float pi = 3.14159265358979323846f;
    if (q < 1.0f) {
        return (3.0f / (2.0f * pi)) * ((2.0f / 3.0f) - q * q + 0.5f * q * q * q);
    } else if (q < 2.0f) {
        float x = 2.0f - q;
        return (1.0f / (4.0f * pi)) * x * x * x;
    }
    return 0.0f;
}

__host__ __device__ inline float ss_kernel_evaluate(float r, float h, float sigma) {
    float q = (r + r) / h;
    return sigma * ss_cubic_function(q);
}

// ---- subdomain that CONTAINS a point in the half-open sense (aabb.rs:220-222 applied to
//      subdomain_grid.cell_aabb, uniform_grid.rs:454-467); used for density ownership
//      (dense_subdomains.rs:567-576).  Corner coordinates of adjacent subdomains are bit-identical, so
//      exactly one subdomain contains a point per axis. ----
__host__ __device__ inline int ss_container_subdomain_axis(const SSDev& P, float x, int d) {
    float normalized = (x - P.gmin[d]) / P.sub_size;
    int s = (int)floorf(normalized);
    float lo = P.gmin[d] + (float)s * P.sub_size;
    float hi = P.gmin[d] + (float)(s + 1) * P.sub_size;
    if (x < lo)
        s -= 1;
    else if (!(x < hi))
        s += 1;
    return s;
}

// ---- absolute search-cell coordinate of x in the neighbourhood-search grid the reference builds for
//      subdomain s (neighborhood_search.rs:370 on the AABB of dense_subdomains.rs:560-565;
//      uniform_grid.rs:189-201 alignment; uniform_grid.rs:444-451 enclosing_cell) ----
__host__ __device__ inline int ss_search_cell_axis(const SSDev& P, int s, float x, int d, int* k_s_out) {
    float amin = P.gmin[d] + (float)s * P.sub_size;
    float mmin = amin - P.margin * 1.5f;
    float kf = floorf(mmin / P.h);
    float aligned = kf * P.h;
    float c = floorf((x - aligned) / P.h);
    if (k_s_out) *k_s_out = (int)kf;
    return (int)kf + (int)c;
}

__host__ __device__ inline uint32_t ss_cell_key(const SSDev& P, int kx, int ky, int kz) {
    return (uint32_t)(((int64_t)(kx - P.kmin[0]) * P.kdim[1] + (ky - P.kmin[1])) * P.kdim[2] + (kz - P.kmin[2]));
}

__host__ __device__ inline bool ss_cell_in_range(const SSDev& P, int kx, int ky, int kz) {
    return kx >= P.kmin[0] && ky >= P.kmin[1] && kz >= P.kmin[2] && kx < P.kmin[0] + P.kdim[0] &&
           ky < P.kmin[1] + P.kdim[1] && kz < P.kmin[2] + P.kdim[2];
}
