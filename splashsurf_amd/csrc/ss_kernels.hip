// ss_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the surface-reconstruction path.
//
// Stage map (reference function -> kernel), citations relative to /root/reference/splashsurf_lib/src/:
//   K0  aabb.rs:28-52 (par_from_points)                                   -> k_aabb_partial/k_aabb_final
//   K0b lib.rs:369-406 (particle AABB filter)                             -> k_inside_flags/k_compact_xyz
//   K1  dense_subdomains.rs:349-494 (decomposition) + neighborhood_search.rs:679-710 (cell map)
//                                                                        -> k_cell_keys, ss_radix_sort_pairs (ss_prims.hip), k_sorted_gather_runs
//   K2  dense_subdomains.rs:496-646 + neighborhood_search.rs:345-438 + density_map.rs:150-186
//                                                                        -> classify scan (SSClassifyIn), k_emit_copies, sort, k_sorted_gather_runs, k_density_sub
//   K3  dense_subdomains.rs:784-847 / :991-1133 (density_grid_loop_scalar / _avx)
//                                                                        -> k_mark_blocks, k_splat_fused, k_select_redo; over-dense blocks: k_splat_certify_big,
//                                                                           k_big_tile_select, k_splat_bounds, k_splat_gather[_large], k_splat_accumulate_list
//   K4  dense_subdomains.rs:1470-1553 (triangulate_cell) classification   -> marching-cubes flag scan (SSMcFlagIn), k_mc_neighbours, k_mc_count
//   K5  same, vertex/triangle emission + dense_subdomains.rs:1603-1749    -> offsets scan (SSMcCountsIn), k_mc_emit
// The scans and the sort are the library's own (ss_prims.h): their input / output functors, i.e. the kernels fused into them, are at the end of this file.
//
// Design: the particles are sorted once by splat cell (aligned with the lattice of level-set blocks) for the level set and
// copied per subdomain, in the reference's own search grids, for the densities.  Every grid point / cell / edge is owned by exactly one
// thread which GATHERS its contributions in ascending original particle index -- the summation order of the reference (sorted
// per-subdomain particle lists, dense_subdomains.rs:476-488) -- so level-set values are bit-identical to the reference's scalar path,
// independent of subdomain or GPU boundaries, with no R atomics anywhere.
#include <algorithm>

#include "ss_device.h"
#include "ss_kernels.h"
#include "ss_prims.h"

// MC table in emitted (winding-flipped) order (mc_table.inc, see tools/gen_mc_table.py), packed at compile time: per case ONE 64-bit word
// -- fifteen 4-bit edge ids (15 = none), the number of triangles in the top nibble -- and the triangle counts alone as nibbles (32 words).
// The count kernel stages 128 bytes, the emit kernel 2 KiB per workgroup (the byte table is 4 KiB: staging it made up most of either
// kernel's memory traffic, 0.74 GB per launch on S10M-tank).
struct SSMcPacked {
    unsigned long long row[256];
    uint32_t ntri4[32];
};
constexpr SSMcPacked ss_make_mc_packed() {
    constexpr int8_t t[256][16] = {
#include "mc_table.inc"
    };
    SSMcPacked p{};
    for (int c = 0; c < 256; ++c) {
        unsigned long long row = 0;
        unsigned ntri = 0;
        for (int j = 0; j < 15; ++j) {
            const int e = t[c][j];
            row |= (unsigned long long)(e < 0 ? 15 : e) << (4 * j);
            if (j % 3 == 0 && e >= 0) ++ntri;
        }
        p.row[c] = row | ((unsigned long long)ntri << 60);
        p.ntri4[c >> 3] |= ntri << (4 * (c & 7));
    }
    return p;
}
__constant__ __attribute__((aligned(16))) SSMcPacked c_mc_packed = ss_make_mc_packed();
// The corners of a cell (uniform_grid.rs:825-834) and local edge -> (origin corner, axis) (uniform_grid.rs:856-869), folded into one
// word: 5 bits per local edge e, (ox << 4) | (oy << 3) | (oz << 2) | axis of the edge's origin corner -- decoded with a shift instead
// of five dependent byte loads per triangle corner
constexpr unsigned long long ss_mc_edge_code(int e) {
    constexpr int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
    constexpr int edge[12][2] = {{0, 0}, {1, 1}, {3, 0}, {0, 1}, {4, 0}, {5, 1}, {7, 0}, {4, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}};
    return (unsigned long long)((corner[edge[e][0]][0] << 4) | (corner[edge[e][0]][1] << 3) | (corner[edge[e][0]][2] << 2) | edge[e][1]);
}
constexpr unsigned long long ss_mc_edge_codes() {
    unsigned long long w = 0;
    for (int e = 0; e < 12; ++e) w |= ss_mc_edge_code(e) << (5 * e);
    return w;
}
#define SS_MC_EDGE_CODES ss_mc_edge_codes()

// =====================================================================================================
// K0: bounding box
// =====================================================================================================
// (partial: SS_AABB_PARTIAL_STRIDE values per block -- min[3], max[3], and 1 if the block saw a coordinate that is not finite: ss_min / ss_max skip a NaN like
// the reference's f32::min / max do (aabb.rs:28-52), so the box alone would not tell, and a NaN coordinate has no cell -- the host refuses such input)
template <class R>
__global__ __launch_bounds__(256) void k_aabb_partial(const R* __restrict__ xyz, uint32_t n, R* __restrict__ partial) {
    __shared__ R s_min[3][256];
    __shared__ R s_max[3][256];
    bool bad = false;
    R mn[3] = {std::numeric_limits<R>::infinity(), std::numeric_limits<R>::infinity(), std::numeric_limits<R>::infinity()}, mx[3] = {-std::numeric_limits<R>::infinity(), -std::numeric_limits<R>::infinity(), -std::numeric_limits<R>::infinity()};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        for (int d = 0; d < 3; ++d) {
            R v = xyz[3 * (size_t)i + d];
            mn[d] = ss_min(mn[d], v);
            mx[d] = ss_max(mx[d], v);
            bad = bad || !(v - v == R(0.0));  // NaN or infinity
        }
    }
    for (int d = 0; d < 3; ++d) {
        s_min[d][threadIdx.x] = mn[d];
        s_max[d][threadIdx.x] = mx[d];
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int d = 0; d < 3; ++d) {
                s_min[d][threadIdx.x] = ss_min(s_min[d][threadIdx.x], s_min[d][threadIdx.x + s]);
                s_max[d][threadIdx.x] = ss_max(s_max[d][threadIdx.x], s_max[d][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int d = 0; d < 3; ++d) {
            partial[blockIdx.x * SS_AABB_PARTIAL_STRIDE + d] = s_min[d][0];
            partial[blockIdx.x * SS_AABB_PARTIAL_STRIDE + 3 + d] = s_max[d][0];
        }
        partial[blockIdx.x * SS_AABB_PARTIAL_STRIDE + 6] = any_bad ? R(1.0) : R(0.0);
    }
}

template <class R>
__global__ __launch_bounds__(256) void k_aabb_final(const R* __restrict__ partial, uint32_t nblocks, R* __restrict__ out6, SSMailSlot mail) {
    __shared__ R s_min[3][256];
    __shared__ R s_max[3][256];
    R mn[3] = {std::numeric_limits<R>::infinity(), std::numeric_limits<R>::infinity(), std::numeric_limits<R>::infinity()}, mx[3] = {-std::numeric_limits<R>::infinity(), -std::numeric_limits<R>::infinity(), -std::numeric_limits<R>::infinity()};
    bool bad = false;
    for (uint32_t i = threadIdx.x; i < nblocks; i += blockDim.x) {
        for (int d = 0; d < 3; ++d) {
            mn[d] = ss_min(mn[d], partial[i * SS_AABB_PARTIAL_STRIDE + d]);
            mx[d] = ss_max(mx[d], partial[i * SS_AABB_PARTIAL_STRIDE + 3 + d]);
        }
        bad = bad || partial[i * SS_AABB_PARTIAL_STRIDE + 6] != R(0.0);
    }
    for (int d = 0; d < 3; ++d) {
        s_min[d][threadIdx.x] = mn[d];
        s_max[d][threadIdx.x] = mx[d];
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int d = 0; d < 3; ++d) {
                s_min[d][threadIdx.x] = ss_min(s_min[d][threadIdx.x], s_min[d][threadIdx.x + s]);
                s_max[d][threadIdx.x] = ss_max(s_max[d][threadIdx.x], s_max[d][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int d = 0; d < 3; ++d) {
            out6[d] = s_min[d][0];
            out6[3 + d] = s_max[d][0];
        }
        ss_mail_post(mail, any_bad ? 1ull : 0ull);  // value: some coordinate is NaN / infinite.  (out6 may be pinned host memory: the release of the post orders the six stores before it)
    }
}

template <class R>
void ss_launch_aabb(const R* d_xyz, uint32_t n, R* d_partial, R* d_out6, SSMailSlot mail, hipStream_t st) {
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_aabb_partial<R>, dim3(blocks), dim3(256), 0, st, d_xyz, n, d_partial);
    hipLaunchKernelGGL(k_aabb_final<R>, dim3(1), dim3(256), 0, st, d_partial, blocks, d_out6, mail);
}

// =====================================================================================================
// K0b: particle AABB filter (lib.rs:369-406; half-open test aabb.rs:220-222)
// =====================================================================================================
template <class R>
__global__ __launch_bounds__(256) void k_inside_flags(const R* __restrict__ xyz, uint32_t n, R a0, R a1, R a2, R b0, R b1,
                                                      R b2, uint8_t* __restrict__ flags8, uint32_t* __restrict__ flags32) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    R x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    bool in = x >= a0 && y >= a1 && z >= a2 && x < b0 && y < b1 && z < b2;
    flags8[i] = in ? 1 : 0;
    flags32[i] = in ? 1u : 0u;
}

template <class R>
__global__ __launch_bounds__(256) void k_compact_xyz(const R* __restrict__ xyz, uint32_t n, const uint32_t* __restrict__ flags32,
                                                     const uint32_t* __restrict__ offsets, R* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags32[i]) return;
    size_t o = offsets[i];
    out[3 * o] = xyz[3 * (size_t)i];
    out[3 * o + 1] = xyz[3 * (size_t)i + 1];
    out[3 * o + 2] = xyz[3 * (size_t)i + 2];
}

template <class R>
void ss_launch_inside_flags(const R* d_xyz, uint32_t n, const R amin[3], const R amax[3], uint8_t* f8, uint32_t* f32, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_inside_flags<R>, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, n, amin[0], amin[1], amin[2], amax[0], amax[1], amax[2], f8,
                       f32);
}
template <class R>
void ss_launch_compact_xyz(const R* d_xyz, uint32_t n, const uint32_t* f32, const uint32_t* offs, R* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_compact_xyz<R>, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, n, f32, offs, out);
}

// =====================================================================================================
// K1: splat-cell keys.  The particles are sorted once by the cells of the splat grid (SSDevT: edge 8 cs / sk, aligned with the lattice
// of level-set blocks, z fastest), so that the particles in reach of a block are a fixed set of contiguous runs (splat_row_cells).
// =====================================================================================================
template <class R>
__device__ inline void ss_particle_cell(const SSDevT<R>& P, R x, R y, R z, int K[3]) {
    K[0] = ss_splat_cell_axis(P, x, 0);
    K[1] = ss_splat_cell_axis(P, y, 1);
    K[2] = ss_splat_cell_axis(P, z, 2);
}

template <class R>
__global__ __launch_bounds__(256) void k_cell_keys(SSDevT<R> P, const R* __restrict__ xyz, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    int K[3];
    ss_particle_cell(P, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], K);
    uint32_t key = ss_cell_key(P, K[0], K[1], K[2]);
    keys[i] = key;
    if (vals) vals[i] = i;  // (null: the sort takes the positions as the values itself)
}

template <class R>
void ss_launch_cell_keys(const SSDevT<R>& P, const R* d_xyz, uint32_t* keys, uint32_t* vals, hipStream_t st) {
    if (!P.n) return;
    hipLaunchKernelGGL(k_cell_keys<R>, dim3((P.n + 255) / 256), dim3(256), 0, st, P, d_xyz, keys, vals);
}
// =====================================================================================================
// K2: per-particle SPH density, organised exactly like the reference (dense_subdomains.rs:496-646):
// every particle is COPIED into each subdomain it belongs to (owner + ghost margins, classification of
// dense_subdomains.rs:1810-1905); the copies of a subdomain are binned in THAT subdomain's own
// neighbourhood-search grid (neighborhood_search.rs:370, cell edge h, origin aligned per subdomain) and
// the copy that lies inside the subdomain's half-open AABB computes
//     rho_i = m * (W(0) + sum_j W(|x_j - x_i|))
// over the 26 adjacent cells in (x,y,z)-lexicographic step order, then the own cell
// (uniform_grid.rs:614-643, neighborhood_search.rs:400-405), ascending original index inside a cell
// (neighborhood_search.rs:692-707).  Using the per-subdomain grids (instead of one global grid) is what
// makes the summation ORDER -- and hence every bit of rho -- identical to the reference even for
// particles that sit exactly on search-cell boundaries.
// =====================================================================================================
// (the literal triple loop of the reference: ghost margins wider than 15 subdomains)
template <class R, class F>
__device__ inline void ss_for_each_member_subdomain_generic(const SSDevT<R>& P, const R p[3], F f) {
    int sub[3];
    R min_corner[3], max_corner[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        sub[d] = (int)ss_floor((p[d] - P.gmin[d]) / P.sub_size);  // uniform_grid.rs:444-451
        if (sub[d] < 0 || sub[d] >= P.ns[d]) return;            // dense_subdomains.rs:1819-1822
        min_corner[d] = P.gmin[d] + (R)sub[d] * P.sub_size;
        max_corner[d] = P.gmin[d] + (R)(sub[d] + 1) * P.sub_size;
    }
    const R dx = P.sub_size;
    const int r = P.sub_radius;  // ceil(margin / dx), dense_subdomains.rs:1827-1832
    for (int i0 = -r; i0 <= r; ++i0)
        for (int j0 = -r; j0 <= r; ++j0)
            for (int k0 = -r; k0 <= r; ++k0) {
                const int steps[3] = {i0, j0, k0};
                bool in_margin = true;
#pragma unroll
                for (int d = 0; d < 3; ++d) {  // dense_subdomains.rs:1844-1856
                    const int step = steps[d];
                    const R off = (R)((step < 0 ? -step : step) - 1);
                    if (step > 0)
                        in_margin = in_margin && (((max_corner[d] + off * dx) - p[d]) < P.margin);
                    else if (step < 0)
                        in_margin = in_margin && ((p[d] - (min_corner[d] - off * dx)) < P.margin);
                }
                if (!in_margin) continue;
                const int nx = sub[0] + i0, ny = sub[1] + j0, nz = sub[2] + k0;
                if (nx < 0 || ny < 0 || nz < 0 || nx >= P.ns[0] || ny >= P.ns[1] || nz >= P.ns[2]) continue;  // :1895-1900
                // multi-GPU shard: only the subdomains this process reconstructs
                if (nx < P.sub_lo[0] || ny < P.sub_lo[1] || nz < P.sub_lo[2] || nx >= P.sub_hi[0] || ny >= P.sub_hi[1] || nz >= P.sub_hi[2]) continue;
                f(nx, ny, nz);
            }
}

// The membership rule of dense_subdomains.rs:1810-1905 -- subdomain sub + (i0, j0, k0) is a member iff, on every axis, the step's
// margin test holds -- is separable: per axis a mask of the valid steps (margin test :1844-1856, grid bounds :1895-1900, this
// process's shard window), and the members are the product of the three sets, visited in the reference's order (i0, j0, k0
// ascending).  A particle has 1.95 member subdomains on average: looping over the 27 (or (2r+1)^3) combinations and testing each
// cost ~400 instructions per particle in the membership count and in k_emit_copies.
template <class R, class F>
__device__ inline void ss_for_each_member_subdomain(const SSDevT<R>& P, const R p[3], F f) {
    int sub[3];
    uint32_t valid[3];
    const R dx = P.sub_size;
    const int r = P.sub_radius;  // ceil(margin / dx), dense_subdomains.rs:1827-1832
    if (r > 15) {  // (wave-uniform) the step masks below have 32 bits
        ss_for_each_member_subdomain_generic(P, p, f);
        return;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        sub[d] = (int)ss_floor((p[d] - P.gmin[d]) / P.sub_size);  // uniform_grid.rs:444-451
        if (sub[d] < 0 || sub[d] >= P.ns[d]) return;            // dense_subdomains.rs:1819-1822
        const R min_corner = P.gmin[d] + (R)sub[d] * P.sub_size;
        const R max_corner = P.gmin[d] + (R)(sub[d] + 1) * P.sub_size;
        auto step_ok = [&](int step) {
            const R off = (R)((step < 0 ? -step : step) - 1);
            bool ok = true;
            if (step > 0)
                ok = ((max_corner + off * dx) - p[d]) < P.margin;  // dense_subdomains.rs:1844-1856
            else if (step < 0)
                ok = (p[d] - (min_corner - off * dx)) < P.margin;
            const int nd = sub[d] + step;
            ok = ok && nd >= 0 && nd < P.ns[d];                      // :1895-1900
            return ok && nd >= P.sub_lo[d] && nd < P.sub_hi[d];      // multi-GPU shard: only the subdomains this process reconstructs
        };
        uint32_t m = 0;
        if (r == 1) {  // (wave-uniform) the margin is narrower than a subdomain: the three steps spelled out, their constants folded
            m = (step_ok(-1) ? 1u : 0u) | (step_ok(0) ? 2u : 0u) | (step_ok(1) ? 4u : 0u);
        } else {
            for (int step = -r; step <= r; ++step) m |= step_ok(step) ? (1u << (step + r)) : 0u;
        }
        valid[d] = m;
    }
    for (uint32_t mi = valid[0]; mi; mi &= mi - 1u) {
        const int nx = sub[0] + (__ffs((int)mi) - 1) - r;
        for (uint32_t mj = valid[1]; mj; mj &= mj - 1u) {
            const int ny = sub[1] + (__ffs((int)mj) - 1) - r;
            for (uint32_t mk = valid[2]; mk; mk &= mk - 1u) f(nx, ny, sub[2] + (__ffs((int)mk) - 1) - r);
        }
    }
}

// cell of x in the neighbourhood-search grid of subdomain index s (per axis); neighborhood_search.rs:370
// applied to the AABB of dense_subdomains.rs:560-565 (uniform_grid.rs:189-201 alignment, :444-451 cell)
template <class R>
__device__ inline int ss_local_search_cell_axis(const SSDevT<R>& P, int s, R x, int d) {
    const R amin = P.gmin[d] + (R)s * P.sub_size;
    const R mmin = amin - P.margin * R(1.5);
    const R aligned = ss_floor(mmin / P.h) * P.h;
    const int c = (int)ss_floor((x - aligned) / P.h);
    return max(0, min(P.sc[d] - 1, c));
}

template <class R>
__global__ __launch_bounds__(256) void k_emit_copies(SSDevT<R> P, const R* __restrict__ xyz, const uint32_t* __restrict__ copy_offset,
                                                     const uint32_t* __restrict__ occ_rank, uint32_t* __restrict__ keys,
                                                     uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const R p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    uint32_t o = copy_offset[i];
    const uint32_t ctot = (uint32_t)(P.sc[0] * P.sc[1] * P.sc[2]);
    ss_for_each_member_subdomain(P, p, [&](int sx, int sy, int sz) {
        const uint32_t occ = occ_rank[((size_t)sx * P.ns[1] + sy) * P.ns[2] + sz];
        const int cx = ss_local_search_cell_axis(P, sx, p[0], 0);
        const int cy = ss_local_search_cell_axis(P, sy, p[1], 1);
        const int cz = ss_local_search_cell_axis(P, sz, p[2], 2);
        const uint32_t key = occ * ctot + (uint32_t)((cx * P.sc[1] + cy) * P.sc[2] + cz);
        keys[o] = key;
        vals[o] = i;
        ++o;
    });
}

// ---- exactly rounded building blocks of W(r) for the density and splat inner loops ---------------------------------
// sqrt: v_sqrt_f32 is accurate to 1 ulp; one residual test against the two neighbouring floats makes it
// correctly rounded (the sequence hipcc emits for ss_sqrt, minus its scaling for inputs below 2^-96, which
// the caller excludes).
__device__ __forceinline__ float ss_sqrt_rn_normal(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1);
    const float sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

template <class R, bool FAST>
__device__ __forceinline__ R ss_div_by_h(R x, R h, R rh) {
    if constexpr (FAST) {
        static_assert(sizeof(R) == 4, "the verified reciprocal division exists for f32 only");
        const float q0 = x * rh;
        const float e = __builtin_fmaf(-q0, h, x);
        return __builtin_fmaf(e, rh, q0);
    } else {
        return x / h;
    }
}

// kernel.rs:71-81 without branches (both polynomial pieces, then select); same association order
template <class R>
__device__ __forceinline__ R ss_cubic_function_sel(R q) {
    const R pi = R(3.14159265358979323846);
    const R fa = (R(3.0) / (R(2.0) * pi)) * ((R(2.0) / R(3.0)) - q * q + R(0.5) * q * q * q);
    const R x = R(2.0) - q;
    const R fb = (R(1.0) / (R(4.0) * pi)) * x * x * x;
    return (q < R(1.0)) ? fa : ((q < R(2.0)) ? fb : R(0.0));
}

// W(sqrt(d2)) exactly as kernel.rs:103-106 evaluates it.
// FAST variant (enabled by the host only for 1e-9 < h < 1e15 and after k_verify_fast_div passed): lean sqrt
// and reciprocal division.  Both are exact for normal-range arguments; for d2 below (2^-14 h)^2 -- the
// only place where v_sqrt_f32 could see a denormal or the division's residual could underflow -- the
// value of q is irrelevant: any q < 2^-13 makes 2/3 - q*q round to 2/3 and 0.5*q^3 vanish, i.e. W == W(0)
// bit for bit, and every path yields such a q there (r is never over-estimated).
template <class R, bool FAST>
__device__ __forceinline__ R ss_kernel_w(R d2, R h, R rh, R sigma) {
    R r;
    if constexpr (FAST)
        r = ss_sqrt_rn_normal(d2);
    else
        r = ss_sqrt(d2);  // generic variant: hipcc's fully guarded, correctly rounded sqrt
    if constexpr (FAST) {
        // q = RN((r + r) / h) = RN(r / (h/2)): halving h and doubling its reciprocal are exact, so the division verified for
        // the divisor h (all significands) serves h/2 as well and the doubling of r is not needed
        const R q = ss_div_by_h<R, true>(r, R(0.5) * h, rh + rh);
        // kernel.rs:71-81.  The outer piece (1 <= q < 2) is evaluated for every lane; the inner piece only if some lane of
        // the wave needs it (a wave-uniform branch): about three quarters of the tile entries a wave visits lie farther
        // than h/2 from all of its 64 points.
        const R pi = R(3.14159265358979323846);
        // q >= 2 (d^2 in [h^2, 1.01 h^2)) must give exactly 0: clamping x = 2 - q at 0 does, since c*0*0*0 == +0.  The
        // clamp to [0, 1] is the subtraction's output modifier (no extra instruction); its upper bound only touches lanes
        // with q < 1, whose value is replaced by the inner piece below.
        R x;
        SS_SUB_CLAMP(x, 2.0, q);
        R f = (R(1.0) / (R(4.0) * pi)) * x * x * x;
        const bool inner = q < R(1.0);
        if (__ballot(inner)) {
            const R fa = (R(3.0) / (R(2.0) * pi)) * ((R(2.0) / R(3.0)) - q * q + R(0.5) * q * q * q);
            f = inner ? fa : f;
        }
        return sigma * f;
    } else {
        const R q = ss_div_by_h<R, false>(r + r, h, rh);
        return sigma * ss_cubic_function_sel<R>(q);
    }
}

// Only ~15 % of the candidates of a particle's 27 cells are neighbours, but in a wave of 64 particles nearly every candidate
// step has SOME lane with a neighbour, so evaluating W inside the candidate loop makes every lane pay for it at every step.
// The loop therefore only computes d^2 and appends it to a per-lane queue in LDS (unconditional store, the fill count
// advances on a hit); W is evaluated for the queued values in lock-step once some lane's queue runs full.  Each lane still
// adds its neighbours' W in candidate order, i.e. in the reference's order.
template <class R> struct SSDensityQueue {
    // 16 KiB of LDS per 256-thread workgroup either way.  Measured on S10M-cube / S1M / S10M-tank (density stage, ms):
    // cap 32: 21.1 / 1.66 / 3.97; 24: 18.1 / 1.49 / 3.73; 16: 17.4 / 1.53 / 3.69; 12: 17.6 / 1.49 / 3.69 -- a deeper queue is
    // fuller when it is flushed but leaves fewer waves per SIMD to cover the dependent chains of the W evaluation.
#ifndef SS_DENSITY_QCAP
#define SS_DENSITY_QCAP 16
#endif
#ifndef SS_DENSITY_CHUNK
#define SS_DENSITY_CHUNK 4
#endif
    static constexpr int cap = sizeof(R) == 4 ? SS_DENSITY_QCAP : 8;
    static constexpr int chunk = SS_DENSITY_CHUNK;  // candidates between two fill checks
};
// MODE 0: densities.  MODE 1: densities + neighbour counts (global_neighborhood_list).
// MODE 2: write the neighbour ids (global particle indices) at nb_ptr[i], in the reference's order
// (dense_subdomains.rs:617-639: the per-subdomain lists remapped to global indices).
template <class R, int MODE, bool FAST>
__global__ __launch_bounds__(256) void k_density_sub(SSDevT<R> P, uint32_t n_copies, const ss_pos<R>* __restrict__ cpos, const uint32_t* __restrict__ cidx,
                                                     const uint32_t* __restrict__ ckey, const uint32_t* __restrict__ cell_start,
                                                     const uint32_t* __restrict__ occ_sub, R* __restrict__ rho,
                                                     uint32_t* __restrict__ nb_count, const unsigned long long* __restrict__ nb_ptr,
                                                     uint32_t* __restrict__ nb_idx, const uint32_t* __restrict__ owned_list,
                                                     const uint32_t* __restrict__ n_owned_dev) {
    constexpr int QC = SSDensityQueue<R>::cap, QD = SSDensityQueue<R>::chunk;
    __shared__ R s_q[(MODE == 2) ? 1 : QC][256];
    const int tid = threadIdx.x;
    // Threads walk the OWNED copies only (owned flags of k_sorted_gather_runs + scan + compaction, in cell order): about half of the copies are
    // ghosts, and a wave of owners and idling ghosts costs as much as a wave of owners.
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_owned_dev) return;
    const uint32_t p = owned_list[t];
    const uint32_t ctot = (uint32_t)(P.sc[0] * P.sc[1] * P.sc[2]);
    const uint32_t key = ckey[p];
    const uint32_t occ = key / ctot, cell = key - occ * ctot;
    const uint32_t flat = occ_sub[occ];
    const int sz = (int)(flat % (uint32_t)P.ns[2]);
    const int sy = (int)((flat / (uint32_t)P.ns[2]) % (uint32_t)P.ns[1]);
    const int sx = (int)(flat / ((uint32_t)P.ns[2] * (uint32_t)P.ns[1]));
    const ss_pos<R> pi = cpos[p];
    (void)sx;
    (void)sy;
    (void)sz;
    const int cz = (int)(cell % (uint32_t)P.sc[2]);
    const int cy = (int)((cell / (uint32_t)P.sc[2]) % (uint32_t)P.sc[1]);
    const int cx = (int)(cell / ((uint32_t)P.sc[2] * (uint32_t)P.sc[1]));
    const uint32_t base = occ * ctot;
    R acc = P.w0;  // density_map.rs:173
    uint32_t nn = 0;
    unsigned long long wr = (MODE == 2) ? nb_ptr[cidx[p]] : 0ull;
    // The 27 cells in the reference's order (step (x,y,z) lexicographic, own cell last) are 11 runs of
    // the cell-sorted copy array, because cells that differ only in z are contiguous: eight (x,y) rows
    // of three cells, then z-1 and z+1 of the own row, then the own cell.  All run bounds are fetched
    // first (independent loads), then the runs are walked.
    uint32_t rb[11], re[11];
    {
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, P.sc[2] - 1);
        int k = 0;
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox)
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy) {
                if (ox == 0 && oy == 0) continue;
                const int nx = cx + ox, ny = cy + oy;
                uint32_t b = 0, e = 0;
                if (nx >= 0 && ny >= 0 && nx < P.sc[0] && ny < P.sc[1]) {
                    const uint32_t row = base + (uint32_t)((nx * P.sc[1] + ny) * P.sc[2]);
                    b = cell_start[row + (uint32_t)z0];
                    e = cell_start[row + (uint32_t)z1 + 1u];
                }
                // rows before the own row come first, rows after it later: slots 0..3 and 6..9
                const int slot = (k < 4) ? k : k + 2;
                rb[slot] = b;
                re[slot] = e;
                ++k;
            }
        const uint32_t row = base + (uint32_t)((cx * P.sc[1] + cy) * P.sc[2]);
        const uint32_t own_b = cell_start[row + (uint32_t)cz], own_e = cell_start[row + (uint32_t)cz + 1u];
        rb[4] = (cz > 0) ? cell_start[row + (uint32_t)cz - 1u] : own_b;  // cell z-1 of the own row
        re[4] = own_b;
        rb[5] = own_e;                                                   // cell z+1 of the own row
        re[5] = (cz + 1 < P.sc[2]) ? cell_start[row + (uint32_t)cz + 2u] : own_e;
        rb[10] = own_b;                                                  // own cell last
        re[10] = own_e;
    }
    const R rh = R(1.0) / P.h;
    uint32_t cnt = 0;  // fill of this lane's queue
    auto flush = [&]() {
        for (uint32_t k = 0; __any(k < cnt); ++k)
            if (k < cnt) acc += ss_kernel_w<R, FAST>(s_q[(MODE == 2) ? 0 : k][tid], P.h, rh, P.sigma);  // density_map.rs:179-180
        nn += cnt;
        cnt = 0;
    };
    // Over-dense input: a search cell holds more particles than a wave has lanes, all 64 lanes sit in ONE cell and walk the same eleven
    // runs.  The candidates are then fetched with wave-uniform addresses (scalar loads, one fetch for the wave instead of one per lane):
    // the kernel is bound by the vector L1 otherwise (S10M-cube: 2 160 candidates per particle).
    bool uniform = false;
    if constexpr (MODE != 2) uniform = __ballot(key != (uint32_t)__builtin_amdgcn_readfirstlane((int)key)) == 0ull;
    if constexpr (MODE != 2) {
      if (uniform) {
#pragma unroll
        for (int run = 0; run < 11; ++run) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)re[run]);
            for (uint32_t q = (uint32_t)__builtin_amdgcn_readfirstlane((int)rb[run]); q < e; q += (uint32_t)QD) {
                if (__any(cnt > (uint32_t)(QC - QD))) flush();
                const ss_pos<R>* cq = cpos + q;  // (wave-uniform)
                ss_pos<R> pj[QD];
#pragma unroll
                for (int j = 0; j < QD; ++j) pj[j] = cq[j];
#pragma unroll
                for (int j = 0; j < QD; ++j) {
                    const uint32_t qq = q + (uint32_t)j;
                    const R dx = pj[j].x - pi.x, dy = pj[j].y - pi.y, dz = pj[j].z - pi.z;
                    const R d2 = dx * dx + dy * dy + dz * dz;
                    bool hit = qq < e && d2 < P.h2;  // neighborhood_search.rs:431
                    if (run == 10) hit = hit && qq != p;
                    s_q[cnt][tid] = d2;
                    cnt += hit ? 1u : 0u;
                }
            }
        }
      }
    }
    if (!uniform) {
#pragma unroll
    for (int run = 0; run < 11; ++run) {
        const uint32_t e = re[run];
        for (uint32_t q = rb[run]; q < e; q += (uint32_t)QD) {
            if (MODE != 2)
                if (__any(cnt > (uint32_t)(QC - QD))) flush();
            // QD candidates per trip, loads issued together from one base address; slots beyond the run are predicated
            // off (they read the next cells' copies -- cpos is padded by QD entries at its end)
            const ss_pos<R>* cq = cpos + q;
            ss_pos<R> pj[QD];
#pragma unroll
            for (int j = 0; j < QD; ++j) pj[j] = cq[j];
#pragma unroll
            for (int j = 0; j < QD; ++j) {
                const uint32_t qq = q + (uint32_t)j;
                const R dx = pj[j].x - pi.x, dy = pj[j].y - pi.y, dz = pj[j].z - pi.z;
                const R d2 = dx * dx + dy * dy + dz * dz;
                bool hit = qq < e && d2 < P.h2;  // neighborhood_search.rs:431
                if (run == 10) hit = hit && qq != p;  // the particle itself sits in its own cell (the last run)
                if (MODE == 2) {
                    if (hit) nb_idx[wr++] = cidx[qq];
                } else {
                    s_q[(MODE == 2) ? 0 : cnt][tid] = d2;
                    cnt += hit ? 1u : 0u;
                }
            }
        }
    }
    }
    if (MODE != 2) flush();
    if (MODE != 2) rho[cidx[p]] = acc * P.mass;  // density_map.rs:182, dense_subdomains.rs:596-614
    if (MODE == 1) nb_count[cidx[p]] = nn;
}

// (x, y, z, V = m / rho) in global-cell order for the splat (v_i of dense_subdomains.rs:832)
template <class R>
__global__ __launch_bounds__(256) void k_make_posvol(SSDevT<R> P, const ss_pos<R>* __restrict__ pos_sorted, const uint32_t* __restrict__ perm,
                                                     const R* __restrict__ rho, ss_real4<R>* __restrict__ posvol, ss_real4<R>* __restrict__ posvol_by_index) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P.n) return;
    const ss_pos<R> a = pos_sorted[p];
    const uint32_t i = perm[p];
    const ss_real4<R> v = ss_make4(a.x, a.y, a.z, P.mass / rho[i]);
    posvol[p] = v;
}

// the same payload in ORIGINAL particle order: the large-tile path (k_splat_gather_large) sorts particle indices only and fetches the payload
// through this copy -- made only when a call has over-dense blocks (a scattered 16-byte store per particle otherwise wasted)
template <class R>
__global__ __launch_bounds__(256) void k_posvol_by_index(uint32_t n, const ss_real4<R>* __restrict__ posvol, const uint32_t* __restrict__ perm, ss_real4<R>* __restrict__ posvol_by_index) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) posvol_by_index[perm[p]] = posvol[p];
}
template <class R>
void ss_launch_posvol_by_index(uint32_t n, const ss_real4<R>* posvol, const uint32_t* perm, ss_real4<R>* posvol_by_index, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_posvol_by_index<R>, dim3((n + 255) / 256), dim3(256), 0, st, n, posvol, perm, posvol_by_index);
}
template void ss_launch_posvol_by_index<float>(uint32_t, const ss_real4<float>*, const uint32_t*, ss_real4<float>*, hipStream_t);
template void ss_launch_posvol_by_index<double>(uint32_t, const ss_real4<double>*, const uint32_t*, ss_real4<double>*, hipStream_t);

template <class R>
void ss_launch_emit_copies(const SSDevT<R>& P, const R* xyz, const uint32_t* copy_offset, const uint32_t* occ_rank, uint32_t* keys, uint32_t* vals,
                           hipStream_t st) {
    if (!P.n) return;
    hipLaunchKernelGGL(k_emit_copies<R>, dim3((P.n + 255) / 256), dim3(256), 0, st, P, xyz, copy_offset, occ_rank, keys, vals);
}
template <class R>
void ss_launch_density_sub(const SSDevT<R>& P, uint32_t n_copies, const ss_pos<R>* cpos, const uint32_t* cidx, const uint32_t* ckey,
                           const uint32_t* cell_start, const uint32_t* occ_sub, R* rho, int mode, uint32_t* nb_count,
                           const unsigned long long* nb_ptr, uint32_t* nb_idx, bool fast_div, const uint32_t* owned_list, const uint32_t* n_owned_dev,
                           uint32_t n_owned_bound, hipStream_t st) {
    if (!n_copies) return;
    const dim3 g((n_owned_bound + 255) / 256), b(256);
    if constexpr (sizeof(R) == 4) {
        if (fast_div && mode != 2) {  // lean exact sqrt and verified reciprocal division inside W (see ss_kernel_w)
            if (mode == 0)
                hipLaunchKernelGGL((k_density_sub<R, 0, true>), g, b, 0, st, P, n_copies, cpos, cidx, ckey, cell_start, occ_sub, rho, nb_count, nb_ptr, nb_idx, owned_list, n_owned_dev);
            else
                hipLaunchKernelGGL((k_density_sub<R, 1, true>), g, b, 0, st, P, n_copies, cpos, cidx, ckey, cell_start, occ_sub, rho, nb_count, nb_ptr, nb_idx, owned_list, n_owned_dev);
            return;
        }
    }
    if (mode == 0)
        hipLaunchKernelGGL((k_density_sub<R, 0, false>), g, b, 0, st, P, n_copies, cpos, cidx, ckey, cell_start, occ_sub, rho, nb_count, nb_ptr, nb_idx, owned_list, n_owned_dev);
    else if (mode == 1)
        hipLaunchKernelGGL((k_density_sub<R, 1, false>), g, b, 0, st, P, n_copies, cpos, cidx, ckey, cell_start, occ_sub, rho, nb_count, nb_ptr, nb_idx, owned_list, n_owned_dev);
    else
        hipLaunchKernelGGL((k_density_sub<R, 2, false>), g, b, 0, st, P, n_copies, cpos, cidx, ckey, cell_start, occ_sub, rho, nb_count, nb_ptr, nb_idx, owned_list, n_owned_dev);
}
template <class R>
void ss_launch_make_posvol(const SSDevT<R>& P, const ss_pos<R>* pos_sorted, const uint32_t* perm, const R* rho, ss_real4<R>* posvol,
                           ss_real4<R>* posvol_by_index, hipStream_t st) {
    if (!P.n) return;
    hipLaunchKernelGGL(k_make_posvol<R>, dim3((P.n + 255) / 256), dim3(256), 0, st, P, pos_sorted, perm, rho, posvol, posvol_by_index);
}

// =====================================================================================================
// K3 prepare: which 8^3-point level-set blocks can receive a contribution?  One thread per splat cell; a non-empty cell marks every
// block whose scan visits it: cell c lies in row (ix, iy) = (cx - sk bx, cy - sk by) of block (bx, by, .) iff 0 <= ix, iy < sn1, and
// the scan of that row covers the cells sk bz + [zlo, zhi] (splat_row_cells: the row's z extent, the same for every block).
// =====================================================================================================
// Row r = ix * sn1 + iy of a block's sn1 x sn1 (x, y) rows of splat cells: the z-range [zlo, zhi] of cells (relative to the block's
// first covering cell) that can hold a particle within reach of the block's points.  In units of cs relative to the block's first
// point the points span [0, 7]^3 and the cells of row (ix, iy) the slab [so + ix se, so + (ix + 1) se] x [so + iy se, ...]: with
// its distance d_xy to [0, 7]^2 only sqrt(rho^2 - d_xy^2) is left along z, which trims the corner rows (a quarter to a third of
// the candidates).  srho is padded (make_device_params), everything here only has to be conservative.
template <class R>
__device__ __forceinline__ bool splat_row_cells_xy(const SSDevT<R>& P, int ix, int iy, int* zlo, int* zhi) {
    const float lx = P.so + (float)ix * P.se, ly = P.so + (float)iy * P.se;
    const float ex = fmaxf(fmaxf(lx - 7.0f, -(lx + P.se)), 0.0f);
    const float ey = fmaxf(fmaxf(ly - 7.0f, -(ly + P.se)), 0.0f);
    const float left = P.srho * P.srho - (ex * ex + ey * ey) * 0.99999f;
    if (left < 0.0f) return false;
    const float rz = __builtin_amdgcn_sqrtf(left) * 1.00001f + 1.0e-4f * P.se;
    const float inv_e = (float)P.sk * 0.125f;
    *zlo = max((int)floorf((-rz - P.so) * inv_e), 0);
    *zhi = min((int)floorf((7.0f + rz - P.so) * inv_e), P.sn1 - 1);
    return *zlo <= *zhi;
}
template <class R>
__device__ __forceinline__ bool splat_row_cells(const SSDevT<R>& P, int r, uint32_t* lo_off, uint32_t* hi_off) {
    // r / sn1 for 0 <= r < 2^14, sn1 < 2^14 (f32 division of small integers, corrected: no integer division)
    int ix = (int)((float)r / (float)P.sn1);
    ix -= (ix * P.sn1 > r) ? 1 : 0;
    ix += ((ix + 1) * P.sn1 <= r) ? 1 : 0;
    const int iy = r - ix * P.sn1;
    int zlo, zhi;
    if (!splat_row_cells_xy<R>(P, ix, iy, &zlo, &zhi)) return false;
    const uint32_t row = (uint32_t)(ix * P.kdim[1] + iy) * (uint32_t)P.kdim[2];
    *lo_off = row + (uint32_t)zlo;
    *hi_off = row + (uint32_t)zhi + 1u;
    return true;
}

// The rows are the same for every block: made once per parameter set, read by k_splat_fused (one 8-byte load per row instead of the
// arithmetic above per block).  (0, 0): the row holds no cell within reach.
template <class R>
__global__ __launch_bounds__(256) void k_splat_row_table(SSDevT<R> P, uint2* __restrict__ tab) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.sn1 * P.sn1) return;
    uint32_t lo_off, hi_off;
    tab[r] = splat_row_cells<R>(P, r, &lo_off, &hi_off) ? make_uint2(lo_off, hi_off) : make_uint2(0u, 0u);
}
template <class R>
void ss_launch_splat_row_table(const SSDevT<R>& P, uint2* tab, hipStream_t st) {
    const int n = P.sn1 * P.sn1;
    hipLaunchKernelGGL(k_splat_row_table<R>, dim3((n + 255) / 256), dim3(256), 0, st, P, tab);
}
template void ss_launch_splat_row_table<float>(const SSDevT<float>&, uint2*, hipStream_t);
template void ss_launch_splat_row_table<double>(const SSDevT<double>&, uint2*, hipStream_t);

template <class R>
__global__ __launch_bounds__(256) void k_mark_blocks(SSDevT<R> P, const uint32_t* __restrict__ cell_start, uint32_t ncells,
                                                     uint32_t* __restrict__ block_flag) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncells) return;
    if (cell_start[c + 1] == cell_start[c]) return;
    int k[3];
    k[2] = (int)(c % (uint32_t)P.kdim[2]) + P.kmin[2];
    k[1] = (int)((c / (uint32_t)P.kdim[2]) % (uint32_t)P.kdim[1]) + P.kmin[1];
    k[0] = (int)(c / ((uint32_t)P.kdim[2] * (uint32_t)P.kdim[1])) + P.kmin[0];
    int blo[3], bhi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        // blocks b with sk b <= k <= sk b + sn1 - 1 (k >= 0: the table starts at the first block's first cell)
        const int lo = k[d] - P.sn1 + 1;
        blo[d] = max(lo > 0 ? (lo + P.sk - 1) / P.sk : 0, P.blk_lo[d]);
        bhi[d] = min(k[d] / P.sk, P.blk_hi[d]);
        if (blo[d] > bhi[d]) return;
    }
    for (int bx = blo[0]; bx <= bhi[0]; ++bx)
        for (int by = blo[1]; by <= bhi[1]; ++by) {
            int zlo, zhi;
            if (!splat_row_cells_xy<R>(P, k[0] - P.sk * bx, k[1] - P.sk * by, &zlo, &zhi)) continue;
            for (int bz = blo[2]; bz <= bhi[2]; ++bz) {
                const int iz = k[2] - P.sk * bz;
                uint32_t* f = block_flag + ss_block_index(P, bx, by, bz);  // (inside [blk_lo, blk_hi], hence inside the table)
                if (iz >= zlo && iz <= zhi && !*f) *f = 1u;             // (a set flag is not written again: the stores of many cells to one word serialise)
            }
        }
}

template <class R>
void ss_launch_mark_blocks(const SSDevT<R>& P, const uint32_t* cell_start, uint32_t ncells, uint32_t* block_flag, hipStream_t st) {
    if (!ncells) return;
    hipLaunchKernelGGL(k_mark_blocks<R>, dim3((ncells + 255) / 256), dim3(256), 0, st, P, cell_start, ncells, block_flag);
}
// =====================================================================================================
// K3: level-set splat in gather form.
//
// Work unit: one active block of 8x8x8 grid points; ONE WAVE per block (k_splat_fused), lane l = point ((l>>4)&3, (l>>2)&3, l&3)
// of the current 4x4x4 sub-block.
//   1. gather: the (x, y) rows of search cells touching the dilated block box are contiguous runs of the cell-sorted particle
//      array (rows trimmed to the sphere's z extent); the wave streams them six batches of 64 candidates at a time (all loads of a
//      group in flight together, splat_wave_scan_grouped), tests every particle against the box spanned by the block's points and
//      keeps the survivors -- payload (x, y, z, V) and particle index -- in LDS, in scan order.
//   2. certify: for each of the eight sub-blocks a LOWER BOUND of the level set from the entries close to the sub-block (any
//      order, cheap arithmetic).  f32: the entries within the near radius of the block's box get a 16-byte record of f16 operands
//      in place of their payload, the near entries of all eight sub-blocks are listed as byte indices in one pass, and every list goes
//      through the MATRIX PIPE 32 entries x 32 points at a time (splat_cert_record / splat_cert_mfma: s u is bilinear in an entry's and a
//      point's vector, the bound C4 u^4 leaves three VALU instructions per pair); f64: phase A tests 64 tile entries at once against the
//      sub-block's box, phase B walks the survivors (splat_accumulate_wave).  If the bound exceeds the threshold at all 64 points the
//      sub-block lies inside the surface and is neither evaluated nor stored.
//   3. evaluate: if sub-blocks remain, the wave ranks the tile by ORIGINAL particle index (splat_sort_tile: an order array, the
//      tile stays where it is) -- the reference's per-point summation order (sorted per-subdomain particle lists,
//      dense_subdomains.rs:476-488) -- and evaluates G += V * W(|x - p|) in the reference's arithmetic
//      (dense_subdomains.rs:828-841 / :1077-1107, the ARITH template parameter).
//   4. Certified sub-blocks with a point next to a grid point outside the surface are evaluated by a second launch over a list
//      (k_select_redo): those are the values marching cubes interpolates with.
// A block with more candidates than a wave holds (SSWaveChunk; over-dense input) takes the ARENA PATH instead: k_splat_bounds (an
// upper bound of the tile size from the row lengths) -> exclusive scan = offsets in one tile arena -> k_splat_gather (one wave per
// block, tiles up to SS_WTILE entries, rank-sorted) / k_splat_gather_large (512 threads per block: scan order up to
// SS_SORT_TILE_MAX entries, beyond that index keys in LDS, bitonic network, several passes over ascending index ranges found by
// bisection, so any input density stays exact) -> k_splat_accumulate_list (512 threads per block, wave w = sub-block w, the tile
// streamed through LDS in chunks, the same certify / order / evaluate steps).
// =====================================================================================================
__global__ __launch_bounds__(256) void k_verify_fast_div(float h, float rh, uint32_t* __restrict__ bad) {
    // all significands of the binade [2^e, 2^(e+1)) that contains h
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
    const uint32_t ebits = __float_as_uint(h) & 0x7F800000u;
    const float x = __uint_as_float(ebits | m);
    const float q_fast = ss_div_by_h<float, true>(x, h, rh);
    const float q_ref = x / h;
    if (__float_as_uint(q_fast) != __float_as_uint(q_ref)) atomicAdd(bad, 1u);
}

void ss_launch_verify_fast_div(float h, float rh, uint32_t* bad, hipStream_t st) {
    hipLaunchKernelGGL(k_verify_fast_div, dim3((1u << 23) / 256), dim3(256), 0, st, h, rh, bad);
}

__device__ __forceinline__ void ss_wave_lds_sync() {
    // LDS operations of one wave complete in order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define SS_WAVE_LIST 66  // survivors of one 64-entry batch per wave (+ 2 slots the read-ahead may touch)
// lower bound of the cubic spline in u = 1 - q^2 (ss_splat_pair, SS_ARITH_BOUND): u^3 (C0 + C1 u^2) <= W(q) / sigma
#define SS_BOUND_C0 0.150818f
// -DSS_PHASE_PROF (tools/build_variant.sh NAME -- -DSS_PHASE_PROF): wave-cycles per phase of k_splat_fused, summed over all waves
// into g_phase_prof (256 rows of 16 counters against atomic contention), read with ss_debug_phase_prof (tools/phase_prof.py)
#ifdef SS_PHASE_PROF
__device__ unsigned long long g_phase_prof[256 * 16];
#define SS_PROF_BEGIN() unsigned long long prof_t0 = __builtin_amdgcn_s_memtime()
#define SS_PROF_MARK(i)                                                                                            \
    do {                                                                                                           \
        const unsigned long long prof_t1 = __builtin_amdgcn_s_memtime();                                           \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_phase_prof[(blockIdx.x & 255u) * 16u + (i)], prof_t1 - prof_t0); \
        prof_t0 = __builtin_amdgcn_s_memtime();                                                                    \
    } while (0)
extern "C" void ss_debug_phase_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[256 * 16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_prof), sizeof(h));
    for (int i = 0; i < 16; ++i) {
        out16[i] = 0;
        for (int r = 0; r < 256; ++r) out16[i] += h[r * 16 + i];
    }
    if (reset) {
        for (auto& v : h) v = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_prof), h, sizeof(h));
    }
}
#else
#define SS_PROF_BEGIN() do { } while (0)
#define SS_PROF_MARK(i) do { } while (0)
#endif
#define SS_BOUND_C1 0.785260f
template <class R, int CAP>
struct SplatShared {
    uint32_t idx[CAP];               // original particle indices of the tile (the sort keys)
    uint32_t row_start[SS_MAX_ROWS];
    uint32_t row_prefix[SS_MAX_ROWS + 1];
    uint32_t wave_tot[8];
    uint32_t count;
};

// Conservative block-level filter of the splat: does the particle lie within reach of the box spanned by the block's grid
// points [plo, phi]?  Same expression as the per-wave test in splat_accumulate_wave on a box that contains every wave's
// sub-block, and all operations involved are monotone under rounding, so whatever a wave accepts passes here as well.
template <class R>
__device__ __forceinline__ R ss_block_box_dist2(const SSDevT<R>& P, const ss_real4<R>& pv, const R plo[3], const R phi[3]) {
    const R ex = ss_max(ss_max(plo[0] - pv.x, pv.x - phi[0]) - P.coord_slack, R(0.0));
    const R ey = ss_max(ss_max(plo[1] - pv.y, pv.y - phi[1]) - P.coord_slack, R(0.0));
    const R ez = ss_max(ss_max(plo[2] - pv.z, pv.z - phi[2]) - P.coord_slack, R(0.0));
    return ex * ex + ey * ey + ez * ez;
}
// The same distance from a box that has the slack folded into its corners (lo - slack, hi + slack; f32): max3(lo' - p, p - hi', 0) per axis, three
// instructions instead of five.  The two forms differ by a rounding of the corner -- an ulp of a coordinate, against a slack of 16 ulp of the LARGEST
// coordinate and reach / near radii padded by 1e-4 relative: every filter built on either stays a superset of what the exact arithmetic lets contribute.
__device__ __forceinline__ float ss_box_dist2_folded(const ss_real4<float>& pv, const float lo[3], const float hi[3]) {
    const float ex = __builtin_fmaxf(__builtin_fmaxf(lo[0] - pv.x, pv.x - hi[0]), 0.0f);  // (v_max3_f32)
    const float ey = __builtin_fmaxf(__builtin_fmaxf(lo[1] - pv.y, pv.y - hi[1]), 0.0f);
    const float ez = __builtin_fmaxf(__builtin_fmaxf(lo[2] - pv.z, pv.z - hi[2]), 0.0f);
    return ex * ex + ey * ey + ez * ez;
}
template <class R>
__device__ __forceinline__ bool ss_within_reach_of_block(const SSDevT<R>& P, const ss_real4<R>& pv, const R plo[3], const R phi[3]) {
    return ss_block_box_dist2<R>(P, pv, plo, phi) <= P.R2;
}

// box of the block's points [plo, phi]; key0 = table index of the first splat cell of the block's covering range: the block scans the
// rows key0 + (ix * kdim[1] + iy) * kdim[2] + [zlo, zhi] of splat_row_cells (the same offsets for every block)
template <class R>
__device__ __forceinline__ uint32_t splat_block_box(const SSDevT<R>& P, const int b3[3], R plo[3], R phi[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int i0 = b3[d] * SS_BLOCK;
        const int i1 = min(i0 + SS_BLOCK - 1, P.np[d] - 1);
        plo[d] = P.gmin[d] + (R)i0 * P.cs;
        phi[d] = P.gmin[d] + (R)i1 * P.cs;
    }
    return ss_cell_key(P, P.sk * b3[0], P.sk * b3[1], P.sk * b3[2]);
}

// ---- one wave visits every particle of the search-cell rows overlapping a block's dilated box -------------------------------
// f(inside, src, id, pv) is called in lock-step for 64 candidates at a time and returns whether the scan goes on (inside = within reach of the block's points, src =
// position in the cell-sorted arrays, id = original particle index if NEED_ID).  Rows are handled 64 at a time (a block
// overlaps more than 64 rows only when the cube size approaches the support radius).
template <class R, bool NEED_ID, class F>
__device__ __forceinline__ void splat_wave_scan(const SSDevT<R>& P, const ss_real4<R>* __restrict__ posvol, const uint32_t* __restrict__ perm,
                                                const uint32_t* __restrict__ cell_start, uint32_t key0, const R plo[3], const R phi[3],
                                                uint32_t* s_row_start, uint32_t* s_row_prefix, int lane, F f) {
    const int nrows = P.sn1 * P.sn1;
    for (int row_base = 0; row_base < nrows; row_base += 64) {
        const int nb = min(64, nrows - row_base);
        uint32_t len = 0;
        if (lane < nb) {
            uint32_t lo_off, hi_off, rb = 0, re = 0;
            if (splat_row_cells<R>(P, row_base + lane, &lo_off, &hi_off)) {
                rb = cell_start[key0 + lo_off];
                re = cell_start[key0 + hi_off];
            }
            s_row_start[lane] = rb;
            len = re - rb;
        }
        uint32_t incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        s_row_prefix[lane] = incl - len;
        const uint32_t total = __shfl(incl, 63);
        ss_wave_lds_sync();
        for (uint32_t q0 = 0; q0 < total; q0 += 64u) {
            const uint32_t q = q0 + (uint32_t)lane;
            bool inside = false;
            uint32_t src = 0, id = 0;
            ss_real4<R> pv = ss_make4(R(0.0), R(0.0), R(0.0), R(0.0));
            if (q < total) {
                int lo = 0, hi = nb - 1;  // last row r with row_prefix[r] <= q
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_row_prefix[mid] <= q)
                        lo = mid;
                    else
                        hi = mid - 1;
                }
                src = s_row_start[lo] + (q - s_row_prefix[lo]);
                pv = posvol[src];
                if (NEED_ID) id = perm[src];
                inside = ss_within_reach_of_block<R>(P, pv, plo, phi);
            }
            if (!f(inside, src, id, pv)) return;  // (wave-uniform)
        }
        ss_wave_lds_sync();  // the next batch overwrites the row tables
    }
}

// Inclusive prefix sum over the wave with DPP row shifts and row broadcasts (no LDS round trips; ss_wave_reduce_to_lane63 has
// the same structure): Hillis-Steele inside the rows of 16, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows 2, 3.
__device__ __forceinline__ uint32_t ss_wave_inclusive_scan(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1, zeros shifted in
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// splat_wave_scan for the fused kernel: the scan is a chain of latencies (row table -> row of a candidate -> its payload), and a
// wave that waits holds one of the SIMD's six wave slots, so the candidates are taken SS_SCAN_GROUP batches of 64 at a time:
// the row look-ups of all batches of a group run interleaved (branch-free bisection over the row prefix table in LDS), then
// all their loads are in flight together, then the batches are handed to f in order: f(d2, src, id, pv) with d2 = the candidate's
// squared box distance to the block's points (<= P.R2: within reach; infinite for the lanes past the end), otherwise as in splat_wave_scan.
#ifndef SS_SCAN_GROUP
#define SS_SCAN_GROUP 4  // (round 6: 4 / 5 / 6 batches in flight give 3.11-3.21 / 3.15-3.19 / 3.23-3.27 ms on S10M-tank: a block's ~216 candidates are one group of four)
#endif
#ifndef SS_FUSED_BAIL
#define SS_FUSED_BAIL 3
#endif
template <class R, bool NEED_ID, class F>
__device__ __forceinline__ void splat_wave_scan_grouped(const SSDevT<R>& P, const ss_real4<R>* __restrict__ posvol, const uint32_t* __restrict__ perm,
                                                        const uint32_t* __restrict__ cell_start, const uint2* __restrict__ row_tab, uint32_t key0, const R plo[3], const R phi[3],
                                                        uint32_t* s_row_start, uint32_t* s_row_prefix, int lane, uint32_t bail_total, uint32_t* bailed, F f) {
    const int nrows = P.sn1 * P.sn1;
    [[maybe_unused]] float flo[3] = {0.0f, 0.0f, 0.0f}, fhi[3] = {0.0f, 0.0f, 0.0f};  // f32: the block's box with the coordinate slack folded in (ss_box_dist2_folded)
    if constexpr (sizeof(R) == 4) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            flo[d] = plo[d] - P.coord_slack;
            fhi[d] = phi[d] + P.coord_slack;
        }
    }
    for (int row_base = 0; row_base < nrows; row_base += 64) {
        const int nb = min(64, nrows - row_base);
        uint32_t len = 0;
        if (lane < nb) {
            uint32_t rb = 0, re = 0;
            const uint2 t = row_tab[row_base + lane];  // (k_splat_row_table)
            if (t.y > t.x) {
                rb = cell_start[key0 + t.x];
                re = cell_start[key0 + t.y];
            }
            s_row_start[lane] = rb;
            len = re - rb;
        }
        const uint32_t incl = ss_wave_inclusive_scan(len);
        s_row_prefix[lane] = (lane < nb) ? incl - len : 0xFFFFFFFFu;  // rows past the end are never chosen by the bisection
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total > bail_total) {  // (wave-uniform) far more candidates in these rows than the caller can hold: it does not want them one by one
            *bailed = total;
            return;
        }
        ss_wave_lds_sync();
        for (uint32_t q0 = 0; q0 < total; q0 += 64u * SS_SCAN_GROUP) {
            uint32_t src[SS_SCAN_GROUP];
#pragma unroll
            for (int j = 0; j < SS_SCAN_GROUP; ++j) {
                const uint32_t q = min(q0 + 64u * (uint32_t)j + (uint32_t)lane, total - 1u);
                int lo = 0;  // last row r with row_prefix[r] <= q
                if (nb > 16) {  // (wave-uniform; a block of the usual grids has 3 x 3 rows: the two widest steps would find nothing)
                    lo += (s_row_prefix[32] <= q) ? 32 : 0;
                    lo += (s_row_prefix[lo + 16] <= q) ? 16 : 0;
                }
#pragma unroll
                for (int step = 8; step > 0; step >>= 1) lo += (s_row_prefix[lo + step] <= q) ? step : 0;
                src[j] = s_row_start[lo] + (q - s_row_prefix[lo]);
            }
            ss_real4<R> pv[SS_SCAN_GROUP];
            uint32_t id[SS_SCAN_GROUP];
#pragma unroll
            for (int j = 0; j < SS_SCAN_GROUP; ++j) {
                pv[j] = posvol[src[j]];
                id[j] = NEED_ID ? perm[src[j]] : 0u;
            }
#pragma unroll
            for (int j = 0; j < SS_SCAN_GROUP; ++j) {
                const uint32_t qj = q0 + 64u * (uint32_t)j;
                if (qj >= total) break;  // (wave-uniform)
                // squared box distance of the candidate to the block's points (infinite beyond the rows' end): f decides what is within reach
                R d2;
                if constexpr (sizeof(R) == 4)
                    d2 = (qj + (uint32_t)lane < total) ? ss_box_dist2_folded(pv[j], flo, fhi) : INFINITY;
                else
                    d2 = (qj + (uint32_t)lane < total) ? ss_block_box_dist2<R>(P, pv[j], plo, phi) : R(INFINITY);
                if (!f(d2, src[j], id[j], pv[j])) return;  // (wave-uniform)
            }
        }
        ss_wave_lds_sync();  // the next batch overwrites the row tables
    }
}

// XCD-aware mapping of groups of four consecutive blocks to 256-thread workgroups (one wave per block): hardware places
// workgroup w on XCD w % 8.  Every XCD gets chunks of SS_XCD_CHUNK consecutive groups of the spatially ordered active list
// (chunk c -> XCD c % 8): neighbouring blocks, which share most of their candidate rows, share an L2, and the cheap and the
// expensive regions of the list (deep inside the fluid / at the surface) spread evenly over the XCDs.
#define SS_XCD_CHUNK 64u
__device__ __forceinline__ uint32_t ss_xcd_chunked_group(uint32_t w) {
    const uint32_t x = w & 7u, s = w >> 3;
    return ((s / SS_XCD_CHUNK) * 8u + x) * SS_XCD_CHUNK + (s % SS_XCD_CHUNK);
}
__device__ __forceinline__ uint32_t ss_xcd_chunked_grid_dev(uint32_t n_groups) {
    const uint32_t span = 8u * SS_XCD_CHUNK;
    return ((n_groups + span - 1u) / span) * span;
}
static inline uint32_t ss_xcd_chunked_grid(uint32_t n_groups) {
    const uint32_t span = 8u * SS_XCD_CHUNK;
    return ((n_groups + span - 1u) / span) * span;
}

__device__ __forceinline__ bool splat_wave_block(uint32_t n_active, uint32_t* logical) {
    *logical = ss_xcd_chunked_group(blockIdx.x) * 4u + (threadIdx.x >> 6);
    return *logical < n_active;
}

// step 1: bound[b] = particles in the (z-trimmed) search-cell rows a block's dilated box overlaps -- an upper bound of its tile
// size that costs two table look-ups per row instead of a distance test per particle.  The exclusive scan of the bounds
// places every block's tile in the arena; only counts[b] <= bound[b] entries of a range are ever written or read.
template <class R>
__global__ __launch_bounds__(256) void k_splat_bounds(SSDevT<R> P, const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ active_xyz, uint32_t n_active,
                                                      const uint32_t* __restrict__ counts, uint32_t* __restrict__ bound) {
    const uint32_t logical = blockIdx.x * blockDim.x + threadIdx.x;
    if (logical > n_active) return;
    uint32_t u = 0;
    // only the blocks k_splat_fused handed on get a tile in the arena (counts: their candidates)
    if (logical < n_active && counts[logical] > (uint32_t)SSWaveChunk<R>::value) {
        const int b3[3] = {(int)active_xyz[3 * (size_t)logical], (int)active_xyz[3 * (size_t)logical + 1], (int)active_xyz[3 * (size_t)logical + 2]};
        const uint32_t key0 = ss_cell_key(P, P.sk * b3[0], P.sk * b3[1], P.sk * b3[2]);
        const int nrows = P.sn1 * P.sn1;
        for (int r = 0; r < nrows; ++r) {
            uint32_t lo_off, hi_off;
            if (splat_row_cells<R>(P, r, &lo_off, &hi_off)) u += cell_start[key0 + hi_off] - cell_start[key0 + lo_off];
        }
    }
    bound[logical] = u;  // entry n_active: 0, so that the exclusive scan ends with the arena size
}


template <class R, int E>
__device__ __forceinline__ void splat_rank_and_write(const uint32_t* s_idx, const uint32_t* s_src, uint32_t count, int lane, const ss_real4<R>* __restrict__ posvol,
                                                     ss_real4<R>* __restrict__ tile) {
    uint32_t my[E], rank[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)(e * 64 + lane);
        my[e] = (i < count) ? s_idx[i] : 0u;
        rank[e] = 0u;
    }
    for (uint32_t k = 0; k < count; ++k) {
        const uint32_t v = s_idx[k];
#pragma unroll
        for (int e = 0; e < E; ++e) rank[e] += (v < my[e]) ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)(e * 64 + lane);
        if (i < count) tile[rank[e]] = posvol[s_src[i]];
    }
}

// step 2, one wave per block: scan, filter, count; a tile of up to SS_WTILE entries is ordered and written right away, a larger one
// (over-dense input) only reports its size and is left to the workgroup-level kernel below
template <class R>
__global__ __launch_bounds__(256) void k_splat_gather(SSDevT<R> P, const ss_real4<R>* __restrict__ posvol, const uint32_t* __restrict__ perm,
                                                      const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ active_xyz, uint32_t n_active,
                                                      const unsigned long long* __restrict__ tile_off, ss_real4<R>* __restrict__ arena, uint32_t* __restrict__ arena_idx,
                                                      uint32_t* __restrict__ counts, uint32_t* __restrict__ large_flag) {
    __shared__ uint32_t s_idx[4][SS_WTILE];
    __shared__ uint32_t s_src[4][SS_WTILE];
    __shared__ uint32_t s_row_start[4][64];
    __shared__ uint32_t s_row_prefix[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t logical;
    if (!splat_wave_block(n_active, &logical)) return;
    if (counts[logical] <= (uint32_t)SSWaveChunk<R>::value) {  // k_splat_fused evaluated this block without a tile in the arena
        if (lane == 0) large_flag[logical] = 0u;
        return;
    }
    const int b3[3] = {(int)active_xyz[3 * (size_t)logical], (int)active_xyz[3 * (size_t)logical + 1], (int)active_xyz[3 * (size_t)logical + 2]};
    R plo[3], phi[3];
    uint32_t count = 0;
    {
        const uint32_t key0 = splat_block_box<R>(P, b3, plo, phi);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        splat_wave_scan<R, true>(P, posvol, perm, cell_start, key0, plo, phi, s_row_start[w], s_row_prefix[w], lane, [&](bool inside, uint32_t src, uint32_t id, const ss_real4<R>&) {
            const unsigned long long m = __ballot(inside);
            const uint32_t pos = count + (uint32_t)__popcll(m & below);
            if (inside && pos < (uint32_t)SS_WTILE) {
                s_idx[w][pos] = id;
                s_src[w][pos] = src;
            }
            count += (uint32_t)__popcll(m);
            return true;
        });
    }
    if (lane == 0) {
        counts[logical] = count;
        large_flag[logical] = (count > (uint32_t)SS_WTILE) ? 1u : 0u;
    }
    if (count > (uint32_t)SS_WTILE) return;
    ss_wave_lds_sync();
    ss_real4<R>* tile = arena + tile_off[logical];
    // (count can be below SSWaveChunk: k_splat_fused hands a block on by the size of its rows, without counting)
    // rank sort by original particle index (unique), payload written in that order.  (An in-register bitonic network -- 28 / 36 /
    // 45 dependent stages through ds_bpermute for 128 / 256 / 512 keys -- issues a third of the instructions but measured
    // slower, 4.1 instead of 3.3 ms on S10M-tank: the rank sort's LDS broadcast reads are independent and pipeline.)
    // Every lane ranks all of its (up to six) elements in ONE pass over the keys: one broadcast read serves them all.
    switch ((count + 63u) >> 6) {
        case 0: break;
        case 1: splat_rank_and_write<R, 1>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
        case 2: splat_rank_and_write<R, 2>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
        case 3: splat_rank_and_write<R, 3>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
        case 4: splat_rank_and_write<R, 4>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
        case 5: splat_rank_and_write<R, 5>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
        default: splat_rank_and_write<R, 6>(s_idx[w], s_src[w], count, lane, posvol, tile); break;
    }
}

// ---- workgroup-level candidate scan of the large-tile gather --------------------------------------------------------------
// exclusive prefix over s.row_prefix[0..nbatch) (lengths in, prefix out), total in row_prefix[nbatch]
template <class S>
__device__ inline void splat_row_prefix(S& s, int nbatch, uint32_t len, int tid) {
    // tid < 256 participate (4 waves); len is this thread's row length (0 beyond nbatch)
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t v = len;
    if (tid < SS_MAX_ROWS) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(v, off);
            if (lane >= off) v += t;
        }
        if (lane == 63) s.wave_tot[wave] = v;
    }
    __syncthreads();
    if (tid < SS_MAX_ROWS) {
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += s.wave_tot[w];
        s.row_prefix[tid] = base + v - len;
        if (tid == nbatch - 1) s.row_prefix[nbatch] = base + v;
    }
    __syncthreads();
}

// Visit every particle of the search cells overlapping the dilated block box; f(idx) is called for particles within reach of
// the block's points.  All 512 threads must call this (contains barriers).
template <class R, class S, class F>
__device__ inline void splat_for_each_candidate(S& s, const SSDevT<R>& P, const ss_real4<R>* __restrict__ posvol,
                                                const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cell_start,
                                                uint32_t key0, const R plo[3], const R phi[3], int tid, F f) {
    const int nrows = P.sn1 * P.sn1;
    for (int row_base = 0; row_base < nrows; row_base += SS_MAX_ROWS) {
        const int nbatch = min(SS_MAX_ROWS, nrows - row_base);
        uint32_t len = 0;
        if (tid < nbatch) {
            uint32_t lo_off, hi_off, b = 0, e = 0;
            if (splat_row_cells<R>(P, row_base + tid, &lo_off, &hi_off)) {
                b = cell_start[key0 + lo_off];
                e = cell_start[key0 + hi_off];
            }
            s.row_start[tid] = b;
            len = e - b;
        }
        __syncthreads();
        splat_row_prefix(s, nbatch, len, tid);
        const uint32_t total = s.row_prefix[nbatch];
        for (uint32_t q = tid; q < total; q += 512) {
            // binary search: last row r with row_prefix[r] <= q
            int lo = 0, hi = nbatch - 1;
            while (lo < hi) {
                int mid = (lo + hi + 1) >> 1;
                if (s.row_prefix[mid] <= q)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            const uint32_t src = s.row_start[lo] + (q - s.row_prefix[lo]);
            const ss_real4<R> pv = posvol[src];
            if (ss_within_reach_of_block<R>(P, pv, plo, phi)) f(perm[src], pv);
        }
        __syncthreads();
    }
}

// step 2, tiles of more than SS_WTILE entries (over-dense input): one 512-thread workgroup per block.  Only the particle
// INDICES (the sort keys) go through LDS, up to CAP per pass; the payload is fetched in sorted order from the copy of
// (x, y, z, V) kept in original particle order.  Tiles beyond CAP are written in several passes over ascending index ranges.
template <class R, int CAP>
__device__ __forceinline__ void splat_gather_large_block(SplatShared<R, CAP>& s, const SSDevT<R>& P, const ss_real4<R>* __restrict__ posvol,
                                                         const ss_real4<R>* __restrict__ posvol_by_index, const uint32_t* __restrict__ perm,
                                                         const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ bxyz, uint32_t expect, ss_real4<R>* __restrict__ tile,
                                                         uint32_t* __restrict__ tile_idx) {
    const int tid = threadIdx.x;
    const int b3[3] = {(int)bxyz[0], (int)bxyz[1], (int)bxyz[2]};
    R plo[3], phi[3];
    const uint32_t key0 = splat_block_box<R>(P, b3, plo, phi);
    if (expect <= (uint32_t)SS_SORT_TILE_MAX) {
        // Left in scan order with the particle indices: the accumulate kernel orders the tile itself if the block needs an exact sum
        // (most blocks inside a body of fluid are certified by the order-independent lower bound).  The payload goes straight from
        // the cell-sorted array the scan reads anyway to the tile -- no second, index-ordered fetch.
        __syncthreads();
        if (tid == 0) s.count = 0;
        __syncthreads();
        splat_for_each_candidate<R>(s, P, posvol, perm, cell_start, key0, plo, phi, tid, [&](uint32_t idx, const ss_real4<R>& pv) {
            const uint32_t pos = atomicAdd(&s.count, 1u);
            if (pos < expect) {
                tile[pos] = pv;
                tile_idx[pos] = idx;
            }
        });
        return;
    }
    long long last = -1;  // particles with original index <= last are already written
    const long long idx_max = (long long)P.n - 1;
    uint32_t written = 0;
    while (written < expect) {
        long long T = idx_max;
        uint32_t total = expect - written;
        if (total > (uint32_t)CAP) {
            // more candidates left than LDS slots: find the largest threshold T with #{last < idx <= T} <= CAP by bisection
            // (the count is monotone in T and grows by at most one per step, so the bracket closes on exactly CAP entries)
            long long lo = last, hi = idx_max;
            while (hi - lo > 1) {
                const long long mid = lo + (hi - lo) / 2;
                __syncthreads();
                if (tid == 0) s.count = 0;
                __syncthreads();
                splat_for_each_candidate<R>(s, P, posvol, perm, cell_start, key0, plo, phi, tid, [&](uint32_t idx, const ss_real4<R>&) {
                    if ((long long)idx > last && (long long)idx <= mid) atomicAdd(&s.count, 1u);
                });
                const uint32_t c = s.count;
                if (c <= (uint32_t)CAP)
                    lo = mid;
                else
                    hi = mid;
            }
            T = lo;
        }
        __syncthreads();
        if (tid == 0) s.count = 0;
        __syncthreads();
        splat_for_each_candidate<R>(s, P, posvol, perm, cell_start, key0, plo, phi, tid, [&](uint32_t idx, const ss_real4<R>&) {
            if ((long long)idx > last && (long long)idx <= T) {
                const uint32_t pos = atomicAdd(&s.count, 1u);
                if (pos < (uint32_t)CAP) s.idx[pos] = idx;
            }
        });
        const int n_tile = (int)min(s.count, (uint32_t)CAP);
        if (n_tile == 0) break;  // cannot happen (the count pass saw `expect` candidates); never spin
        if (n_tile <= 512) {
            if (tid < n_tile) {
                const uint32_t my_idx = s.idx[tid];
                uint32_t rank = 0;
                for (int k = 0; k < n_tile; ++k) rank += (s.idx[k] < my_idx) ? 1u : 0u;
                tile[written + rank] = posvol_by_index[my_idx];
            }
        } else {
            int m = 1024;
            while (m < n_tile) m <<= 1;
            for (int e = n_tile + tid; e < m; e += 512) s.idx[e] = 0xFFFFFFFFu;
            __syncthreads();
            // bitonic network, one workgroup barrier per stage.  (Running the stages with distance j <= 64 wave-synchronously
            // needs 15 instead of 66 barriers for 2048 keys but measured slower: the dependent LDS round trips of one wave no
            // longer overlap.)
            for (int k = 2; k <= m; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (m >> 1); t += 512) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const int l = i | j;
                        const bool up = (i & k) == 0;
                        const uint32_t a = s.idx[i], c = s.idx[l];
                        if ((a > c) == up) {
                            s.idx[i] = c;
                            s.idx[l] = a;
                        }
                    }
                    __syncthreads();
                }
            for (int e = tid; e < n_tile; e += 512) tile[written + (uint32_t)e] = posvol_by_index[s.idx[e]];
        }
        written += (uint32_t)n_tile;
        last = T;
        __syncthreads();
    }
}

template <class R, int CAP>
__global__ __launch_bounds__(512) void k_splat_gather_large(SSDevT<R> P, const ss_real4<R>* __restrict__ posvol, const ss_real4<R>* __restrict__ posvol_by_index,
                                                            const uint32_t* __restrict__ perm, const uint32_t* __restrict__ cell_start,
                                                            const uint32_t* __restrict__ active_xyz, const uint32_t* __restrict__ large_list,
                                                            const uint32_t* __restrict__ n_large_dev, const uint32_t* __restrict__ counts,
                                                            const unsigned long long* __restrict__ tile_off, ss_real4<R>* __restrict__ arena,
                                                            uint32_t* __restrict__ arena_idx) {
    __shared__ SplatShared<R, CAP> s;
    // the list lives on the device (the host never waits for its length; an empty list costs one trivial launch); persistent
    // workgroups walk it, every XCD a contiguous range of the spatially ordered list
    const uint32_t n_large = *n_large_dev;
    const uint32_t per_xcd = (n_large + 7u) / 8u, xcd = blockIdx.x & 7u, stride = gridDim.x >> 3;
    for (uint32_t j = blockIdx.x >> 3; j < per_xcd; j += stride) {
        const uint32_t it = xcd * per_xcd + j;
        if (it < n_large) {
            const uint32_t logical = large_list[it];
            splat_gather_large_block<R, CAP>(s, P, posvol, posvol_by_index, perm, cell_start, active_xyz + 3 * (size_t)logical, counts[logical], arena + tile_off[logical],
                                             arena_idx + tile_off[logical]);
        }
        __syncthreads();
    }
}

// ---- arithmetic of one (tile entry, grid point) pair ------------------------------------------------------------------------
// CubicSplineKernelAvxF32::evaluate (kernel.rs:341-377), one lane of it
__device__ __forceinline__ float ss_kernel_w_avx(const SSDevT<float>& P, float r) {
    const float q = r * P.avx_inv_h;
    // v = max(1 - q, 0): the clamp to [0, 1] is the subtraction's output modifier; q >= 0, so the upper bound never acts
    float v;
    SS_SUB_CLAMP(v, 1.0, q);
    const float v2 = v * v;
    const float v3 = v2 * v;
    float w = v3 * P.avx_sigma2;  // outer piece
    const bool inner = q <= 0.5f;
    if (__ballot(inner)) {  // wave-uniform: most entries a wave visits are farther than h/2 from all of its points
        SS_ASM_NOTE("inner spline piece");  // keeps the compiler from if-converting this block into five always-executed VALU ops
        float ri = __builtin_fmaf(-v, P.avx_sigma6, P.avx_sigma);
        ri = __builtin_fmaf(v2, P.avx_sigma12, ri);
        ri = __builtin_fmaf(-v3, P.avx_sigma6, ri);
        w = inner ? ri : w;
    }
    return w;
}

template <class R, int ARITH>
__device__ __forceinline__ R ss_splat_pair(const SSDevT<R>& P, R rh, const ss_real4<R>& e, R px, R py, R pz, R acc) {
    const R dx = e.x - px, dy = e.y - py, dz = e.z - pz;  // p_i - point, :828 / :1072-1074
    if constexpr (ARITH >= SS_ARITH_SIMD) {
        static_assert(sizeof(R) == 4, "the reference's SIMD loop exists for f32 only (dense_subdomains.rs:1413-1415)");
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));  // :1077-1080
        if constexpr (ARITH == SS_ARITH_BOUND) {
            // lower-bound pass, in units of sigma (the caller scales the sum once).  Entry and point arrive in units of h relative to
            // the sub-block's centre (splat_accumulate_wave), so d2 = q^2, and with u = max(1 - q^2, 0)
            //     u^3 (c0 + c1 u^2) <= W / sigma = min(1 - 6 q^2 + 6 q^3, 2 (1 - q)^3)   on [0, 1], and = 0 beyond
            // (c0, c1 = the linear program "largest integral of g q^2 subject to g <= W" for the basis (u^3, u^5), scaled by 1 - 1e-4;
            // tests/test_oracle.py checks the inequality on 2e5 points, where that scaling exceeds the grid's Lipschitz error).  It
            // carries 96 % of the kernel's mass (the former v^2 min(2 v, 1), v = 1 - q: 90 %) and needs no square root -- six
            // full-rate instructions after d2 instead of five and v_sqrt_f32, a quarter-rate instruction: no reach test (u = 0
            // beyond h), no second piece, no EXEC bookkeeping.
            float u;
            SS_SUB_CLAMP(u, 1.0, d2);
            const float u2 = u * u;
            const float poly = __builtin_fmaf(u2, SS_BOUND_C1, SS_BOUND_C0);
            acc = __builtin_fmaf((u2 * u) * poly, e.w, acc);
        } else if (d2 < P.h2) {                                                     // :1083
            float r;
            if constexpr (ARITH == SS_ARITH_SIMD)
                r = ss_sqrt(d2);
            else if constexpr (ARITH == SS_ARITH_SIMD_LEAN)
                r = ss_sqrt_rn_normal(d2);  // below (2^-25 h)^2 any r rounds v = 1 - q to 1: the value of W does not depend on it
            else
                r = __builtin_amdgcn_sqrtf(d2);
            acc = __builtin_fmaf(ss_kernel_w_avx(P, r), e.w, acc);  // :1101-1107
        }
    } else {
        const R d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < P.H2) {  // :831
            acc += e.w * ss_kernel_w<R, ARITH == SS_ARITH_FAST>(d2, P.h, rh, P.sigma);  // :832-841
        }
    }
    return acc;
}

// Accumulation of one wave's 4^3 sub-block over an index-ordered chunk of the tile in LDS (dense_subdomains.rs:817-841).
// Phase A tests 64 tile entries at once against the sub-block's box; the survivors are compacted IN ORDER into the wave's
// own list `wl` (SS_WAVE_LIST entries), which phase B then walks front to back with a plain counter -- the scalar unit is
// shared by the four SIMDs of a CU and walking a 64-bit survivor mask cost 14 scalar instructions per entry.
// `r2_filter` = squared reach of the sub-block filter: P.R2 for the exact sum (every entry that can contribute), P.R2near for
// the classification pass of splat_accumulate_block (only the entries close to the sub-block).
// `premask` (optional): per tile entry, bit s set <=> the entry passes the filter of sub-block s (splat_near_masks); then the
// box test is not repeated here.  `n_visited` (optional) is advanced by the number of entries that passed the filter.
// `order` (optional): the tile is visited as pay[order[0]], pay[order[1]], ... instead of front to back.
template <class R, int ARITH>
__device__ __forceinline__ R splat_accumulate_wave(const SSDevT<R>& P, const ss_real4<R>* pay, ss_real4<R>* wl, int n_tile, int lane, R px, R py, R pz,
                                                   const R slo[3], const R shi[3], R r2_filter, R acc, const uint8_t* premask = nullptr, int premask_bit = 0,
                                                   int* n_visited = nullptr, const uint8_t* order = nullptr) {
    const R rh = R(1.0) / P.h;
    // lower-bound pass: positions in units of h relative to the sub-block's centre (differences of nearby numbers are exact, the
    // scaling costs a relative 2^-24: nothing against the 1e-4 margin of thr_inside) -- saves the multiplication by 1/h per pair
    R cx = R(0.0), cy = R(0.0), cz = R(0.0);
    if constexpr (ARITH == SS_ARITH_BOUND) {
        cx = R(0.5) * (slo[0] + shi[0]);
        cy = R(0.5) * (slo[1] + shi[1]);
        cz = R(0.5) * (slo[2] + shi[2]);
        px = (px - cx) * P.avx_inv_h;
        py = (py - cy) * P.avx_inv_h;
        pz = (pz - cz) * P.avx_inv_h;
    }
    for (int base = 0; base < n_tile; base += 64) {
        const int c = base + lane;
        bool pass = false;
        ss_real4<R> pv = ss_make4(R(0.0), R(0.0), R(0.0), R(0.0));
        if (premask) {
            if (c < n_tile && ((premask[c] >> premask_bit) & 1u)) {
                pass = true;
                pv = pay[c];
            }
        } else if (c < n_tile) {
            pv = pay[order ? (int)order[c] : c];  // (order: the tile's entries in ascending particle index, splat_sort_tile)
            const R ex = ss_max(ss_max(slo[0] - pv.x, pv.x - shi[0]) - P.coord_slack, R(0.0));
            const R ey = ss_max(ss_max(slo[1] - pv.y, pv.y - shi[1]) - P.coord_slack, R(0.0));
            const R ez = ss_max(ss_max(slo[2] - pv.z, pv.z - shi[2]) - P.coord_slack, R(0.0));
            pass = (ex * ex + ey * ey + ez * ez) <= r2_filter;
        }
        const unsigned long long wmask = __ballot(pass);
        if (wmask) {
            const int cnt = __popcll(wmask);
            if (n_visited) *n_visited += cnt;
            ss_wave_lds_sync();  // the previous batch's reads of wl are done
            if constexpr (ARITH == SS_ARITH_BOUND) pv = ss_make4((pv.x - cx) * P.avx_inv_h, (pv.y - cy) * P.avx_inv_h, (pv.z - cz) * P.avx_inv_h, pv.w);
            if (pass) wl[__builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u))] = pv;
            // the walk below takes two entries per trip: an odd list ends with an entry out of everybody's reach and of volume 0
            // (its term is skipped by the reach test, and +0 in the lower-bound pass)
            if (lane == 0) wl[cnt] = ss_make4(R(1.0e18), R(1.0e18), R(1.0e18), R(0.0));
            ss_wave_lds_sync();
            // Entries are fetched with a wave-uniform LDS read (broadcast) one iteration ahead of their use, ping-ponging between
            // two register sets (no loop-carried copies).  Reads past the end stay inside wl and are not used.  (Measured
            // alternatives on S10M-tank, first pass 6.8 ms: survivors kept one per lane in VGPRs and broadcast with four
            // v_readlane_b32 9.7 ms, every second entry that way 8.2 ms; entries fetched from the tile in global memory through
            // the scalar cache, s_load_dwordx4 one entry ahead, 11.6 ms.)
            ss_real4<R> ea = wl[0];
            for (int k = 0; k < cnt; k += 2) {
                const ss_real4<R> eb = wl[k + 1];
                acc = ss_splat_pair<R, ARITH>(P, rh, ea, px, py, pz, acc);
                ea = wl[k + 2];
                acc = ss_splat_pair<R, ARITH>(P, rh, eb, px, py, pz, acc);
            }
        }
    }
    return acc;
}

// Wave-wide min / max of an f32 ending in lane 63, with DPP row shifts and row broadcasts (12 VALU for both against ~50 for
// the ds_bpermute shuffles of __shfl_xor): after row_shr 1, 2, 4, 8 lane 15 of every row holds its row's result,
// row_bcast15 folds rows 0 -> 1 and 2 -> 3, row_bcast31 folds lane 31 into rows 2 and 3.  Lanes without a source keep `old`.
template <bool IS_MAX>
__device__ __forceinline__ float ss_wave_reduce_to_lane63(float v) {
    auto step = [](float x, int ctrl_sel) -> float {
        const int b = __float_as_int(x);
        int t;
        switch (ctrl_sel) {
            case 0: t = __builtin_amdgcn_update_dpp(b, b, 0x111, 0xF, 0xF, false); break;  // row_shr:1
            case 1: t = __builtin_amdgcn_update_dpp(b, b, 0x112, 0xF, 0xF, false); break;  // row_shr:2
            case 2: t = __builtin_amdgcn_update_dpp(b, b, 0x114, 0xF, 0xF, false); break;  // row_shr:4
            case 3: t = __builtin_amdgcn_update_dpp(b, b, 0x118, 0xF, 0xF, false); break;  // row_shr:8
            case 4: t = __builtin_amdgcn_update_dpp(b, b, 0x142, 0xA, 0xF, false); break;  // row_bcast:15 into rows 1 and 3
            default: t = __builtin_amdgcn_update_dpp(b, b, 0x143, 0xC, 0xF, false); break; // row_bcast:31 into rows 2 and 3
        }
        const float y = __int_as_float(t);
        return IS_MAX ? fmaxf(x, y) : fminf(x, y);
    };
#pragma unroll
    for (int i = 0; i < 6; ++i) v = step(v, i);
    return v;
}

// Which of the six faces of a 4^3 sub-block (lane l <-> point ((l>>4)&3, (l>>2)&3, l&3)) hold a point that is NOT inside the
// surface (marching cubes: inside <=> value > threshold, dense_subdomains.rs:1482).  bit f: 0 -x, 1 +x, 2 -y, 3 +y, 4 -z, 5 +z.
__device__ __forceinline__ uint32_t splat_face_bits(unsigned long long outside) {
    return ((outside & 0x000000000000FFFFull) ? 1u : 0u) | ((outside & 0xFFFF000000000000ull) ? 2u : 0u) | ((outside & 0x000F000F000F000Full) ? 4u : 0u) |
           ((outside & 0xF000F000F000F000ull) ? 8u : 0u) | ((outside & 0x1111111111111111ull) ? 16u : 0u) | ((outside & 0x8888888888888888ull) ? 32u : 0u);
}

// step 3
template <class R>
struct SplatAccShared {
    ss_real4<R> pay[SS_WTILE];
    ss_real4<R> wl[8][SS_WAVE_LIST];
    R red[16];
    uint32_t trunc;
    uint32_t face[8];
    uint32_t skey[SS_SORT_TILE_MAX];  // particle indices of an unordered tile (sort keys) ...
    uint16_t spos[SS_SORT_TILE_MAX];  // ... and the positions of its entries, ordered by index after splat_order_tile
};

// Orders an unordered tile (SS_WTILE < n <= SS_SORT_TILE_MAX entries, written by k_splat_gather_large in scan order): bitonic
// network over (particle index, position) pairs in LDS, one workgroup barrier per stage; afterwards sh.spos[k] is the position in
// the tile of the entry with the k-th smallest index.  All 512 threads.
template <class R>
__device__ inline void splat_order_tile(SplatAccShared<R>& sh, const uint32_t* __restrict__ tile_idx, int n_tile, int tid) {
    int m = 512;
    while (m < n_tile) m <<= 1;
    for (int e = tid; e < m; e += 512) {
        sh.skey[e] = e < n_tile ? tile_idx[e] : 0xFFFFFFFFu;
        sh.spos[e] = (uint16_t)e;
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m >> 1); t += 512) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = (i & k) == 0;
                const uint32_t a = sh.skey[i], c = sh.skey[l];
                if ((a > c) == up) {
                    sh.skey[i] = c;
                    sh.skey[l] = a;
                    const uint16_t pa = sh.spos[i];
                    sh.spos[i] = sh.spos[l];
                    sh.spos[l] = pa;
                }
            }
            __syncthreads();
        }
}

template <class R, int ARITH, bool EARLY>
__device__ __forceinline__ void splat_accumulate_block(SplatAccShared<R>& sh, const SSDevT<R>& P, uint32_t logical, const ss_real4<R>* __restrict__ arena,
                                                       const uint32_t* __restrict__ arena_idx, const unsigned long long* __restrict__ tile_off, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ active_xyz, R* __restrict__ G, ss_real2<R>* __restrict__ blk_minmax,
                                                       uint32_t* __restrict__ trunc, unsigned long long* __restrict__ facebits, uint32_t wave_mask, bool write_faces) {
    // wave_mask: the sub-blocks to evaluate (second pass: the certified ones marching cubes reads); the others keep their values
    // write_faces (!EARLY only): the face bits of the block are written as in the EARLY pass (the exact pass that follows k_splat_certify_big)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tile = (int)counts[logical];
    const ss_real4<R>* tile = arena + tile_off[logical];
    const bool unordered = n_tile > SS_WTILE && n_tile <= SS_SORT_TILE_MAX;  // k_splat_gather_large left this tile in scan order
    ss_real4<R> nxt = ss_make4(R(0.0), R(0.0), R(0.0), R(0.0));
    if (tid < min(n_tile, SS_WTILE)) nxt = tile[tid];  // first chunk in flight while the coordinates are set up
    if (tid == 0) sh.trunc = 0u;
    const int bx = (int)active_xyz[3 * (size_t)logical], by = (int)active_xyz[3 * (size_t)logical + 1], bz = (int)active_xyz[3 * (size_t)logical + 2];
    // this wave's sub-block and this lane's grid point
    const int g0[3] = {bx * SS_BLOCK + ((wave >> 2) & 1) * 4, by * SS_BLOCK + ((wave >> 1) & 1) * 4, bz * SS_BLOCK + (wave & 1) * 4};
    const int gl[3] = {g0[0] + ((lane >> 4) & 3), g0[1] + ((lane >> 2) & 3), g0[2] + (lane & 3)};
    const bool wave_selected = ((wave_mask >> wave) & 1u) != 0u;
    const uint32_t certified_before = (!EARLY && wave_mask != 0xFFu) ? trunc[logical] : 0u;
    const bool wave_valid = wave_selected && g0[0] < P.np[0] && g0[1] < P.np[1] && g0[2] < P.np[2];
    const bool point_valid = gl[0] < P.np[0] && gl[1] < P.np[1] && gl[2] < P.np[2];
    // global point coordinates: uniform_grid.rs:418-425 on the GLOBAL grid (dense_subdomains.rs:817-826); the SIMD loop of the
    // reference forms z with one fma (:1069), x and y like the scalar loop (:1113-1114)
    const R px = P.gmin[0] + (R)gl[0] * P.cs;
    const R py = P.gmin[1] + (R)gl[1] * P.cs;
    R pz;
    if constexpr (ARITH >= SS_ARITH_SIMD)
        pz = __builtin_fmaf((R)gl[2], P.cs, P.gmin[2]);
    else
        pz = P.gmin[2] + (R)gl[2] * P.cs;
    R slo[3], shi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        slo[d] = P.gmin[d] + (R)g0[d] * P.cs;
        shi[d] = P.gmin[d] + (R)min(g0[d] + 3, P.np[d] - 1) * P.cs;
    }
    R acc = R(0.0);  // levelset_grid.fill(0), dense_subdomains.rs:1390
    bool done = false;
    int n_near = 0;  // entries the classification pass visited
    // The tile streams through LDS in chunks of SS_WTILE entries (one chunk for all but over-dense blocks), up to twice:
    //  pass 0 (EARLY only), classification: the sum over the entries CLOSE to the sub-block only (box distance <= 0.60 h, about a
    //   quarter of the tile, but most of every point's kernel mass).  Every term is >= 0, so it bounds the level set from below
    //   whatever the order and whatever the arithmetic -- f32 jobs use the cheapest variant (fma, v_sqrt_f32; within ~1e-6 relative
    //   of every other mode's terms, far inside the margin of thr_inside).  If it exceeds the threshold at all 64 points, the
    //   sub-block lies inside the fluid: marching cubes only needs that fact, unless the block is next to a sign change, in which
    //   case the second kernel pass (k_splat_accumulate_list) evaluates it in full.
    //  pass 1, the exact sum in the reference's order, for the waves pass 0 did not certify.
    constexpr int CLS = (sizeof(R) == 4) ? SS_ARITH_BOUND : ARITH;
    const int n_chunks = (n_tile + SS_WTILE - 1) / SS_WTILE;
    for (int pass = EARLY ? 0 : 1; pass < 2; ++pass) {
        if (pass == 1 && EARLY) {
            if constexpr (CLS == SS_ARITH_BOUND) acc *= P.avx_sigma;  // the bound pass sums in units of sigma
            // the margin of thr_inside covers the rounding of the terms; a sum of n of them adds up to n 2^-24 relative (over-dense
            // tiles hold thousands of near entries)
            const R thr = P.thr_inside + ((R)n_near * R(1.2e-7)) * P.thr_inside;
            done = wave_valid ? (__ballot(acc > thr || !point_valid) == ~0ull) : true;
            // (a single chunk stays in LDS: its waves go on independently, no workgroup barrier between the passes)
            if (n_chunks > 1 && !__syncthreads_or(done ? 0 : 1)) break;  // every sub-block of this block is certified: no second stream
            if (!done) acc = R(0.0);
        }
        const bool by_position = pass == 1 && unordered;  // the exact sum walks an unordered tile through its sorted positions
        if (by_position) splat_order_tile<R>(sh, arena_idx + tile_off[logical], n_tile, tid);
        if (pass == 1 && (by_position || (EARLY && n_chunks > 1)) && tid < min(n_tile, SS_WTILE))
            nxt = by_position ? tile[sh.spos[tid]] : tile[tid];  // the tile is streamed (again)
        for (int c0 = 0; c0 < n_tile; c0 += SS_WTILE) {
            const int nc = min(SS_WTILE, n_tile - c0);
            if (n_chunks > 1 || pass == (EARLY ? 0 : 1)) {  // a single chunk stays in LDS between the passes
                if (tid < nc) sh.pay[tid] = nxt;
                __syncthreads();
                if (c0 + SS_WTILE + tid < n_tile && tid < SS_WTILE)  // next chunk in flight during the arithmetic
                    nxt = by_position ? tile[sh.spos[c0 + SS_WTILE + tid]] : tile[c0 + SS_WTILE + tid];
            }
            if (wave_valid && !done) {
                if (pass == 0)
                    acc = splat_accumulate_wave<R, CLS>(P, sh.pay, sh.wl[wave], nc, lane, px, py, pz, slo, shi, P.R2near, acc, nullptr, 0, &n_near);
                else
                    acc = splat_accumulate_wave<R, ARITH>(P, sh.pay, sh.wl[wave], nc, lane, px, py, pz, slo, shi, P.R2, acc);
            }
            if (n_chunks > 1) __syncthreads();  // pay is overwritten by the next chunk
        }
    }
    if (EARLY && done && wave_valid && lane == 0) atomicOr(&sh.trunc, 1u << wave);  // bit mask of the certified sub-blocks of this block
    // store: block-local layout SS_BLOCK_OFFSET (sub-block after sub-block); the value of dense_subdomains.rs:839
    const int lx = ((wave >> 2) & 1) * 4 + ((lane >> 4) & 3);
    const int ly = ((wave >> 1) & 1) * 4 + ((lane >> 2) & 3);
    const int lz = (wave & 1) * 4 + (lane & 3);
    R* gp = G + (size_t)logical * SS_BLOCK_POINTS + (size_t)SS_BLOCK_OFFSET(lx, ly, lz);
    R val;
    if (wave_selected) {
        if (EARLY && done && wave_valid) {
            val = P.thr_inside;  // certified: nothing is stored, marching cubes takes the fact from the block's mask (mc_load_tile)
        } else {
            val = point_valid ? acc : R(0.0);
            *gp = val;
        }
    } else {
        // second pass: a sub-block that is not re-evaluated keeps its value (it still enters the block's min / max); the values of
        // a sub-block the first pass certified were never stored
        val = ((certified_before >> wave) & 1u) ? P.thr_inside : *gp;
    }
    if (EARLY || write_faces) {
        const unsigned long long outside = __ballot(point_valid && !(val > P.threshold));
        if (lane == 0) sh.face[wave] = splat_face_bits(outside);
    }
    // block-wide min/max of the level-set values (points outside the grid count as 0 = "outside"), used to skip marching cubes
    // on blocks that cannot contain the iso-surface; a truncated wave reports values that are all above the threshold, like
    // its complete values would be
    R mn = val, mx = val;
    if constexpr (sizeof(R) == 4) {
        mn = ss_wave_reduce_to_lane63<false>(mn);
        mx = ss_wave_reduce_to_lane63<true>(mx);
        if (lane == 63) {
            sh.red[wave] = mn;
            sh.red[8 + wave] = mx;
        }
    } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn = ss_min(mn, __shfl_xor(mn, off));
            mx = ss_max(mx, __shfl_xor(mx, off));
        }
        if (lane == 0) {
            sh.red[wave] = mn;
            sh.red[8 + wave] = mx;
        }
    }
    __syncthreads();
    if (tid == 0) {
        mn = sh.red[0];
        mx = sh.red[8];
        for (int q = 1; q < 8; ++q) {
            mn = ss_min(mn, sh.red[q]);
            mx = ss_max(mx, sh.red[8 + q]);
        }
        blk_minmax[logical] = ss_make2(mn, mx);
        trunc[logical] = EARLY ? sh.trunc : (wave_mask == 0xFFu ? 0u : (trunc[logical] & ~wave_mask));
        if (EARLY || write_faces) {
            unsigned long long fb = 0;
            for (int q = 0; q < 8; ++q) fb |= (unsigned long long)sh.face[q] << (6 * q);
            facebits[logical] = fb;
        }
    }
}

// Classification filter of all eight sub-blocks of a block at once, for one tile entry: bit s of the result <=> the entry's box
// distance to sub-block s = (sx << 2) | (sy << 1) | sz is <= sqrt(r2).  The distance separates per axis, two intervals each:
// 6 one-dimensional distances and 12 additions instead of 8 x (3 distances + 2 additions).
template <class R>
__device__ __forceinline__ uint32_t splat_near_masks(const SSDevT<R>& P, const ss_real4<R>& pv, const R lo[3][2], const R hi[3][2], R r2) {
    R e2[3][2];
    const R p[3] = {pv.x, pv.y, pv.z};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const R e = ss_max(ss_max(lo[d][h] - p[d], p[d] - hi[d][h]) - P.coord_slack, R(0.0));
            e2[d][h] = e * e;
        }
    uint32_t m = 0;
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) m |= ((e2[0][(sb >> 2) & 1] + e2[1][(sb >> 1) & 1]) + e2[2][sb & 1] <= r2) ? (1u << sb) : 0u;
    return m;
}

typedef _Float16 ss_half2v __attribute__((ext_vector_type(2)));

// ---- the certificate on the MATRIX PIPE (f32, one wave per block; round 6) -----------------------------------------------------
// The bound  W(q) / sigma >= C4 u^4,  u = max(1 - q^2, 0):  for q >= 1/2 the ratio 2 (1 - q)^3 / (1 - q^2)^4 = 2 / ((1 - q) (1 + q)^4) has its
// minimum 0.762939... at q = 3/5, the inner piece stays above it (tests/test_oracle.py checks the inequality); it carries 90.2 % of the
// kernel's mass (the polynomial u^3 (c0 + c1 u^2) of rounds 3-5 and of the arena kernel's classification, ss_splat_pair: 95.9 %).  It is HOMOGENEOUS in u, so a particle's weight folds
// into the argument,  V sigma C4 u^4 = (s u)^4  with  s = (C4 sigma V)^(1/4),  and
//     s u = s (1 - |p|^2) + 2 s p . x - s |x|^2        (p: entry, x: grid point; relative to the block's centre, in units of h)
// is BILINEAR in a vector of the entry and a vector of the point: ONE v_mfma_f32_32x32x8_f16 evaluates it for 32 entries x 32 points,
//     slot           0          1          2        3     |    4        5       6      7
//     entry  (A)  (s a)_hi   (s a)_lo    2 s px    -s     |  2 s py   2 s pz    -s     -s          a = 1 - |p|^2 - eps
//     point  (B)     1          1          x       x^2    |    y        z      y^2    z^2
// (lanes 0-31 hold slots 0-3 of row / column `lane`, lanes 32-63 slots 4-7 of row / column `lane - 32`; D: column = lane & 31, row =
// (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); layout and rates: tools/ubench/mfma_cert.hip), and THREE VALU instructions per (entry, point)
// pair are left -- in fact two (splat_cert_term4: d |d| clamped to [0, 1] = max(d, 0)^2 in one instruction, then an fmac of the square with itself) --
// against the eleven of the list walk of rounds 3-5 (f16 records, v_fma_mix_f32): ~120 cycles per 1024 pairs per SIMD instead of ~480 (mfma_cert.hip; the matrix
// pipe's 32 cycles do not overlap the VALU's on this chip, they add).
// STILL A LOWER BOUND: the products of f16 operands are exact and summed in f32; against the f32 values an operand rounded to nearest is off by
// 2^-11 relative, so  |P~ x~ - 2 s p x| <= 2 s |p| |x| 2^-10 (1 + 2^-11)  per axis and  |s~ (x^2)~ - s x^2| <= s x^2 2^-10 (1 + 2^-11);
// (s a)_hi + (s a)_lo = s a up to 2^-22.  With xm >= |x| per axis (the block's points: 3.5 cs / h)
//     eps = 2^-10 (1 + 2^-10) (2 xm (|px| + |py| + |pz|) + 3 xm^2) + 3e-5        (P.cert_e1 (|px| + |py| + |pz|) + P.cert_e0)
// taken off a makes the computed value <= s u for every point of the block; the 3e-5 cover the f32 roundings of the operands'
// own computation, the accumulation inside the instruction (eight products of magnitude < 10) and f16 subnormals.  s itself is a
// factor of the whole term: its rounding (v_sqrt_f32 twice, <= 3 ulp) is inside P.cert_vscale = C4 sigma (1 - 2e-5).
typedef _Float16 ss_half4v __attribute__((ext_vector_type(4)));
typedef float ss_float16v __attribute__((ext_vector_type(16)));
#define SS_CERT_C4 0.76293f
#ifndef SS_CERT_DUAL
#define SS_CERT_DUAL 0
#endif
#ifndef SS_ABLATE
#define SS_ABLATE 0  // 1 / 2 / 3: k_splat_fused stops after the scan / after the records and lists / before the exact sums (instruction counts per phase)
#endif
#ifndef SS_CERT_POOL
#define SS_CERT_POOL 512  // bytes of index lists of one block: 64 per sub-block (two tiles of 32 rows)
#endif
__device__ __forceinline__ uint32_t ss_pack_f16(float lo, float hi) {
    const ss_half2v v = {(_Float16)lo, (_Float16)hi};  // (round to nearest even)
    return __builtin_bit_cast(uint32_t, v);
}
// the 16-byte record of a tile entry: slots 0-3 in (x, y), slots 4-7 in (z, w)
// (e1, e0: the slack constants of the frame the coordinates are relative to -- P.cert_e1 / _e0 for a block's centre, P.cert_e1s / _e0s for a sub-block's)
__device__ __forceinline__ uint4 splat_cert_record(const SSDevT<float>& P, const ss_real4<float>& pv, float cx, float cy, float cz, float e1, float e0) {
    const float px = (pv.x - cx) * P.avx_inv_h, py = (pv.y - cy) * P.avx_inv_h, pz = (pv.z - cz) * P.avx_inv_h;
    const float s = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(pv.w * P.cert_vscale));
    const float eps = __builtin_fmaf(e1, (__builtin_fabsf(px) + __builtin_fabsf(py)) + __builtin_fabsf(pz), e0);
    const float a = ((1.0f - eps) - px * px) - (py * py + pz * pz);
    const float sa = s * a;
    const _Float16 sa_hi = (_Float16)sa;
    const float sa_lo = sa - (float)sa_hi;
    const float s2 = s + s;
    const ss_half2v w0 = {sa_hi, (_Float16)sa_lo};
    return make_uint4(__builtin_bit_cast(uint32_t, w0), ss_pack_f16(s2 * px, -s), ss_pack_f16(s2 * py, s2 * pz), ss_pack_f16(-s, -s));
}
#define SS_CERT_DUMMY make_uint4(0x0000E3D0u, 0u, 0u, 0u)  // (s a)_hi = -1000: below zero at every point
// max(d, 0)^4 of four outputs added to acc, TWO instructions each: d |d| with the result clamped to [0, 1] is max(d, 0)^2 -- a negative d gives
// -d^2 -> 0, and d = s u <= s < 1 (s^4 = C4 sigma V <= C4: V W(0) <= 1 for every particle) never meets the upper bound --, written as
// fmed3(d |d|, 0, 1), which hipcc folds into v_mul_f32_e64 v, d, |d| clamp; then the fmac of the square with itself.  (Plain C++ on purpose: the
// wait states between the MFMA and the first reader of its results are inserted for the compiler's own instructions only.)
__device__ __forceinline__ float splat_cert_term4(float d0, float d1, float d2, float d3, float acc) {
    auto t = [](float d, float a) {
        const float m2 = __builtin_amdgcn_fmed3f(d * __builtin_fabsf(d), 0.0f, 1.0f);
        return __builtin_fmaf(m2, m2, a);
    };
    return t(d3, t(d2, t(d1, t(d0, acc))));
}
// one 32 x 32 tile: rows = entries a (A operand, this lane's half of its row's record), columns = points b; returns acc + this lane's part of
// sum_rows max(s u, 0)^4 for its column (rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); `rows` > 0 live rows, the others hold the dummy
__device__ __forceinline__ ss_float16v splat_cert_mfma(uint2 a, uint32_t b0, uint32_t b1) {
    const ss_float16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_bit_cast(ss_half4v, a), __builtin_bit_cast(ss_half4v, make_uint2(b0, b1)), z, 0, 0, 0);
}
__device__ __forceinline__ float splat_cert_reduce(const ss_float16v& d, int rows, float acc) {
    acc = splat_cert_term4(d[0], d[1], d[2], d[3], acc);
    if (rows > 8) acc = splat_cert_term4(d[4], d[5], d[6], d[7], acc);      // (wave-uniform: whole groups of eight rows beyond the list are skipped)
    if (rows > 16) acc = splat_cert_term4(d[8], d[9], d[10], d[11], acc);
    if (rows > 24) acc = splat_cert_term4(d[12], d[13], d[14], d[15], acc);
    return acc;
}

// ---- one WAVE per block -----------------------------------------------------------------------------------------------------
// Blocks whose tile fits one wave's LDS chunk (all blocks of ordinary inputs: ~140 entries at the reference's default spacing)
// are evaluated by a single wave that walks the eight 4^3 sub-blocks one after the other.  Against the workgroup-per-block
// kernel above the work per (entry, point) pair is the same, but a CU keeps ~32 independent blocks in flight instead of 4
// (the loads of the next block's tile hide behind other blocks' arithmetic), there is no workgroup barrier, the tile is
// fetched once per block by one wave, block-uniform values live in SGPRs and the min / max reduction runs once per block.

template <class R>
struct SplatAccWaveShared {
    // the tile in scan order: payload (x, y, z, V); f32 first pass: overwritten entry by entry with the certificate's 16-byte records
    // (splat_cert_record; [CH] = the dummy record the lists are padded with) and fetched again by the blocks that go on to exact sums
    ss_real4<R> pay[SSWaveChunk<R>::value + 1];
    // one 64-entry batch of full records (exact sums); before that the row tables of the scan and -- f32 certificate -- the index lists
    // of the near entries of all eight sub-blocks (SS_CERT_POOL bytes)
    ss_real4<R> wl[SS_WAVE_LIST];
    uint8_t near[SSWaveChunk<R>::value];  // f64: per tile entry the sub-blocks whose classification pass visits it; after splat_sort_tile: the order
    uint32_t idx[SSWaveChunk<R>::value];  // positions of the tile entries in the cell-sorted arrays; for the exact sums: their particle indices (splat_sort_tile)
};
static_assert(sizeof(SplatAccWaveShared<float>::wl) >= SS_CERT_POOL, "the index lists of the certificate live in the list buffer");

// Orders the tile by original particle index (unique keys) WITHOUT moving it: rank sort, every lane ranks its entries in one pass
// over the keys (one broadcast read serves them all), then order[rank] = position goes into the mask array (free once the
// classification is over), and the exact sums read the tile through it (splat_accumulate_wave).  The exact sum needs this order
// (dense_subdomains.rs:817-841 visits the particles in index order); the lower-bound pass does not.  (Moving the payload to its
// rank instead kept twelve registers alive across the ranking loop -- spilled to scratch -- or needed a second copy of the tile
// in LDS, which costs a wave of occupancy.)
template <class R>
__device__ __forceinline__ void splat_sort_tile(SplatAccWaveShared<R>& sh, const uint32_t* __restrict__ tile_idx, int n_tile, int lane) {
    constexpr int E = SSWaveChunk<R>::value / 64;
    static_assert(SSWaveChunk<R>::value <= 256, "positions are stored as bytes");
    if (tile_idx) {  // (the fused kernel has the indices in LDS already)
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (lane + 64 * e < n_tile) sh.idx[lane + 64 * e] = tile_idx[lane + 64 * e];
        ss_wave_lds_sync();
    }
    uint32_t my[E], rank[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane + 64 * e;
        my[e] = (i < n_tile) ? sh.idx[i] : 0u;
        rank[e] = 0u;
    }
    for (int k = 0; k < n_tile; ++k) {
        const uint32_t v = sh.idx[k];
#pragma unroll
        for (int e = 0; e < E; ++e) rank[e] += (v < my[e]) ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (lane + 64 * e < n_tile) sh.near[rank[e]] = (uint8_t)(lane + 64 * e);
    ss_wave_lds_sync();
}

// The tile is in LDS: sh.pay[0, n_tile) payload in scan order, sh.idx its positions in the cell-sorted arrays; f32 first pass: sh.near[0, n_near_block)
// the positions in the tile of the entries within the near radius of the block's box (k_splat_fused).
template <class R, int ARITH, bool EARLY>
__device__ __forceinline__ void splat_accumulate_block_wave(SplatAccWaveShared<R>& sh, const SSDevT<R>& P, uint32_t logical, int n_tile, int n_near_block, const ss_real4<R>* __restrict__ posvol,
                                                            const uint32_t* __restrict__ perm,
                                                            const uint32_t* __restrict__ active_xyz, R* __restrict__ G, ss_real2<R>* __restrict__ blk_minmax,
                                                            uint32_t* __restrict__ trunc, unsigned long long* __restrict__ facebits, uint32_t wave_mask) {
    constexpr int CH = SSWaveChunk<R>::value;
    static_assert(CH % 64 == 0, "the tile is staged in whole batches of 64 entries");
    constexpr bool CERT = EARLY && sizeof(R) == 4;  // f32: the certificate on the matrix pipe; f64: the reference's own arithmetic on the near entries
    const int lane = threadIdx.x & 63;
    SS_PROF_BEGIN();
    const int bx = __builtin_amdgcn_readfirstlane((int)active_xyz[3 * (size_t)logical]);
    const int by = __builtin_amdgcn_readfirstlane((int)active_xyz[3 * (size_t)logical + 1]);
    const int bz = __builtin_amdgcn_readfirstlane((int)active_xyz[3 * (size_t)logical + 2]);
    const int ox = (lane >> 4) & 3, oy = (lane >> 2) & 3, oz = lane & 3;
    R* gblock = G + (size_t)logical * SS_BLOCK_POINTS + (size_t)SS_BLOCK_OFFSET(ox, oy, oz);  // + 64 sb: SS_BLOCK_OFFSET
    // per axis and half of the block (h = 0, 1): the sub-block's box [lo, hi] and this lane's point coordinate -- global point
    // coordinates as in uniform_grid.rs:418-425 on the GLOBAL grid (dense_subdomains.rs:817-826); the SIMD loop of the reference
    // forms z with one fma (:1069), x and y like the scalar loop (:1113-1114)
    R lo[3][2], hi[3][2], pc[3][2];
    bool sub_ok[3][2], pt_ok[3][2];
    {
        const int b3[3] = {bx, by, bz}, o3[3] = {ox, oy, oz};
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = b3[d] * SS_BLOCK + 4 * h;
                lo[d][h] = P.gmin[d] + (R)g * P.cs;
                hi[d][h] = P.gmin[d] + (R)min(g + 3, P.np[d] - 1) * P.cs;
                if (d == 2 && ARITH >= SS_ARITH_SIMD)
                    pc[d][h] = __builtin_fmaf((R)(g + o3[d]), P.cs, P.gmin[d]);
                else
                    pc[d][h] = P.gmin[d] + (R)(g + o3[d]) * P.cs;
                sub_ok[d][h] = g < P.np[d];
                pt_ok[d][h] = g + o3[d] < P.np[d];
            }
    }
    // ---- f32 first pass: records and index lists of the certificate (see splat_cert_record) ----
    // The payload of every entry within the near radius of the BLOCK's box is replaced IN PLACE by its 16-byte record (coordinates relative
    // to the block's centre); the near entries of all eight sub-blocks are listed in ONE pass over those entries as byte indices into the tile: list sb = bytes [64 sb, 64 sb + 64) of the
    // pool, padded with the dummy's index.  The near test of a sub-block (box distance <= R_near, separable: six one-dimensional
    // distances per entry) goes straight into its ballot -- no mask word per entry -- and the ballot's prefix count places the entry.
    // cnt[sb]: entries near sub-block sb.  A sub-block with more near entries than its 64 slots (two tiles; fine grids list ~25, a grid of cube size h / 2
    // over a hundred) walks the BLOCK's near list instead (sh.near, padded to whole tiles below): every entry bounds the level set from below, near or not --
    // the near lists only keep the tiles few --, so the longer walk certifies at least what the sub-block's own list would.
    [[maybe_unused]] int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] uint32_t rb0[4], rb1[4];
    if constexpr (CERT) {
        const float bcx = lo[0][0] + 3.5f * P.cs, bcy = lo[1][0] + 3.5f * P.cs, bcz = lo[2][0] + 3.5f * P.cs;
        float flo[3][2], fhi[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                flo[d][h] = lo[d][h] - P.coord_slack;
                fhi[d][h] = hi[d][h] + P.coord_slack;
            }
        uint4* recs = reinterpret_cast<uint4*>(sh.pay);
        uint8_t* pool = reinterpret_cast<uint8_t*>(sh.wl);
        {
            uint32_t* pool32 = reinterpret_cast<uint32_t*>(sh.wl);
            pool32[lane] = 0x01010101u * (uint32_t)CH;  // padding: the dummy's index
            pool32[lane + 64] = 0x01010101u * (uint32_t)CH;
            if (lane == 0) recs[CH] = SS_CERT_DUMMY;
        }
#pragma unroll
        for (int k = 0; k < CH / 64; ++k) {
            if (64 * k >= n_near_block) break;  // (wave-uniform; S10M-tank: 63 of a tile's 142 entries, one trip)
            const bool valid = lane + 64 * k < n_near_block;
            const int c = valid ? (int)sh.near[lane + 64 * k] : 0;
            const ss_real4<R> pv = sh.pay[c];
            float e2[3][2];
            const float p3[3] = {pv.x, pv.y, pv.z};
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float e = __builtin_fmaxf(__builtin_fmaxf(flo[d][h] - p3[d], p3[d] - fhi[d][h]), 0.0f);  // (the slack folded into the corners: v_max3_f32)
                    e2[d][h] = e * e;
                }
            if (valid) recs[c] = splat_cert_record(P, pv, bcx, bcy, bcz, P.cert_e1, P.cert_e0);
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) {
                const bool bit = valid && ((e2[0][(sb >> 2) & 1] + e2[1][(sb >> 1) & 1]) + e2[2][sb & 1] <= P.R2near);
                const unsigned long long m = __ballot(bit);
                const int pos = cnt[sb] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (bit && pos < 64) pool[64 * sb + pos] = (uint8_t)c;
                cnt[sb] += __popcll(m);
            }
        }
        if (((cnt[0] > 64) | (cnt[1] > 64) | (cnt[2] > 64) | (cnt[3] > 64)) | ((cnt[4] > 64) | (cnt[5] > 64) | (cnt[6] > 64) | (cnt[7] > 64))) {  // (wave-uniform)
            // some sub-block walks the block's near list: pad it to whole tiles of 32 rows with the dummy's index (positions below CH: the array's own)
            const int i = n_near_block + (lane & 31);
            if (lane < 32 && i < ((n_near_block + 31) & ~31)) sh.near[i] = (uint8_t)CH;
        }
        // the B operands: lanes 0-31 hold the x content of their column for its four x positions q = 2 sx + g (g: tile = points
        // [32 g, 32 g + 32) of the sub-block), lanes 32-63 the (y, z) content of their column for q = 2 sy + sz
        const bool lo_half = lane < 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gx = bx * SS_BLOCK + 4 * (q >> 1) + 2 * (q & 1) + ((lane >> 4) & 1);
            const int gy = by * SS_BLOCK + 4 * (q >> 1) + oy, gz = bz * SS_BLOCK + 4 * (q & 1) + oz;
            const float x = ((P.gmin[0] + (float)gx * P.cs) - bcx) * P.avx_inv_h;
            const float y = ((P.gmin[1] + (float)gy * P.cs) - bcy) * P.avx_inv_h;
            const float zc = (ARITH >= SS_ARITH_SIMD) ? __builtin_fmaf((float)gz, P.cs, P.gmin[2]) : P.gmin[2] + (float)gz * P.cs;
            const float z = (zc - bcz) * P.avx_inv_h;
            rb0[q] = ss_pack_f16(lo_half ? 1.0f : y, lo_half ? 1.0f : z);
            rb1[q] = ss_pack_f16(lo_half ? x : y * y, lo_half ? x * x : z * z);
        }
        ss_wave_lds_sync();
    } else if constexpr (EARLY) {
#pragma unroll
        for (int k = 0; k < CH / 64; ++k)
            if (lane + 64 * k < n_tile) sh.near[lane + 64 * k] = (uint8_t)splat_near_masks<R>(P, sh.pay[lane + 64 * k], lo, hi, P.R2near);
        ss_wave_lds_sync();
    }
    SS_PROF_MARK(1);  // block set-up, near masks, records and lists of the certificate
#if SS_ABLATE == 2
    if (n_tile < 100000) return;
#endif
    // second pass: the sub-blocks the first pass certified (their values were never stored, see below)
    const uint32_t certified_before = (!EARLY && wave_mask != 0xFFu) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)trunc[logical]) : 0u;
    R mn = R(INFINITY), mx = -R(INFINITY);
    uint32_t certified = 0, need = 0;
    unsigned long long faces = 0;
    // first the sub-blocks that need no exact sum: not selected (second pass), outside the grid, or certified by the lower bound
    // (certificate: the A operand -- this lane's half of its row's record -- of a sub-block's first tile is fetched one sub-block ahead:
    // index, then record, are two dependent LDS round trips that would otherwise head every sub-block's chain)
    // (list_of(sb): where sub-block sb's list starts -- its 64 slots of the pool, or the block's near list if it outgrew them; wave-uniform)
    [[maybe_unused]] auto list_of = [&](int sb_) -> const uint8_t* { return cnt[sb_] > 64 ? sh.near : reinterpret_cast<const uint8_t*>(sh.wl) + 64 * sb_; };
    [[maybe_unused]] auto cert_row = [&](const uint8_t* lst, int base) -> uint2 {
        const uint32_t e = lst[base + (lane & 31)];
        return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(sh.pay) + e * 16u + (uint32_t)(lane >> 5) * 8u);
    };
    [[maybe_unused]] uint2 arow_ahead = make_uint2(0u, 0u);
    if constexpr (CERT) arow_ahead = cert_row(list_of(0), 0);
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) {
        const int sx = (sb >> 2) & 1, sy = (sb >> 1) & 1, sz = sb & 1;
        [[maybe_unused]] const uint2 arow_first = arow_ahead;
        if constexpr (CERT)
            if (sb < 7) arow_ahead = cert_row(list_of(sb + 1), 0);
        const bool point_valid = pt_ok[0][sx] && pt_ok[1][sy] && pt_ok[2][sz];
        R* gp = gblock + 64 * sb;
        R val;
        if (!((wave_mask >> sb) & 1u)) {
            // second pass: a sub-block that is not re-evaluated keeps its value (it still enters the block's min / max)
            val = ((certified_before >> sb) & 1u) ? P.thr_inside : *gp;
        } else if (!(sub_ok[0][sx] && sub_ok[1][sy] && sub_ok[2][sz])) {
            val = R(0.0);  // points outside the grid count as 0 = "outside"
            *gp = val;
        } else {
            bool done = false;
            if constexpr (EARLY) {  // classification: lower bound from the entries close to the sub-block, in any order
                int n_near = 0;
                R acc;
                if constexpr (CERT) {
                    SS_PROF_MARK(5);  // (classification: everything but the tiles)
                    n_near = cnt[sb];
                    // tiles g = 0, 1 of this sub-block: columns = its points [32 g, 32 g + 32); lanes 0-31 take the x content of position
                    // 2 sx + g, lanes 32-63 the (y, z) content of 2 sy + sz (rb0 is the same constant in all lanes 0-31)
                    const int j = 2 * sy + sz;
                    const bool lo_half = lane < 32;
                    const uint32_t b1g0 = (2 * sx == j) ? rb1[j] : (lo_half ? rb1[2 * sx] : rb1[j]);
                    const uint32_t b1g1 = (2 * sx + 1 == j) ? rb1[j] : (lo_half ? rb1[2 * sx + 1] : rb1[j]);
                    float a0 = 0.0f, a1 = 0.0f;
                    const int n_walk = n_near <= 64 ? n_near : n_near_block;  // (a list that outgrew its slots: the block's near list)
                    const uint8_t* lst = list_of(sb);
                    for (int base = 0; base < n_walk; base += 32) {  // (one trip unless the list has more than 32 entries)
                        const uint2 arow = base == 0 ? arow_first : cert_row(lst, base);
                        const int rows = n_walk - base;
#if SS_CERT_DUAL  // both tiles on the matrix pipe before either is consumed (16 registers more)
                        const ss_float16v d0 = splat_cert_mfma(arow, rb0[j], b1g0), d1 = splat_cert_mfma(arow, rb0[j], b1g1);
                        a0 = splat_cert_reduce(d0, rows, a0);
                        a1 = splat_cert_reduce(d1, rows, a1);
#else
                        a0 = splat_cert_reduce(splat_cert_mfma(arow, rb0[j], b1g0), rows, a0);
                        a1 = splat_cert_reduce(splat_cert_mfma(arow, rb0[j], b1g1), rows, a1);
#endif
                    }
                    // lane l < 32 holds in a0 its part of point l, lane l + 32 the other rows' part of point l (a1: point 32 + l): one
                    // exchange of the halves puts both parts of a lane's OWN point (lane = point of the sub-block) into that lane
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
                    acc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                    SS_PROF_MARK(6);  // tiles of the certificate
                } else {
                    const R px = pc[0][sx], py = pc[1][sy], pz = pc[2][sz];
                    const R slo[3] = {lo[0][sx], lo[1][sy], lo[2][sz]};
                    const R shi[3] = {hi[0][sx], hi[1][sy], hi[2][sz]};
                    acc = splat_accumulate_wave<R, ARITH>(P, sh.pay, sh.wl, n_tile, lane, px, py, pz, slo, shi, P.R2near, R(0.0), sh.near, sb, &n_near);
                }
                // the margin of thr_inside covers the rounding of the terms; a sum of n of them adds up to n 2^-24 relative (the certificate's
                // lists hold at most 64 entries, the block's near list a tile's CH: two constants serve them all)
                const R thr = CERT ? P.thr_inside + (R(n_near <= 64 ? 64 : CH) * R(1.2e-7)) * P.thr_inside : P.thr_inside + ((R)n_near * R(1.2e-7)) * P.thr_inside;
                done = __ballot(acc > thr || !point_valid) == ~0ull;
            }
            if (!done) {
                need |= 1u << sb;
                continue;
            }
            // A certified sub-block stores nothing: marching cubes takes "some value above the threshold" for its points from
            // the block's mask (mc_load_tile), and the second pass writes the sub-blocks whose values are really read.
            certified |= 1u << sb;
            val = P.thr_inside;
            if (P.thr_inside > P.threshold) continue;  // (always, but for thresholds <= 0) every point is inside: no face bits; min / max: after the loop
        }
        mn = ss_min(mn, val);
        mx = ss_max(mx, val);
        if constexpr (EARLY) faces |= (unsigned long long)splat_face_bits(__ballot(point_valid && !(val > P.threshold))) << (6 * sb);
    }
    if (certified && P.thr_inside > P.threshold) {  // the certified sub-blocks' stand-in value enters the block's min / max once
        mn = ss_min(mn, P.thr_inside);
        mx = ss_max(mx, P.thr_inside);
    }
    SS_PROF_MARK(2);  // classification of the eight sub-blocks
#if SS_ABLATE == 3
    need = 0;
#endif
    if (need) {
        // the exact sums want the payload (the certificate overwrote it) and the particle indices: both by the entries' positions in the
        // cell-sorted arrays, rows this block has just read
        ss_wave_lds_sync();
#pragma unroll
        for (int k = 0; k < CH / 64; ++k)
            if (lane + 64 * k < n_tile) {
                const uint32_t src = sh.idx[lane + 64 * k];
                if constexpr (CERT) sh.pay[lane + 64 * k] = posvol[src];
                sh.idx[lane + 64 * k] = perm[src];
            }
        ss_wave_lds_sync();
#if SS_ABLATE != 4  // (4: the exact sums without the tile sort -- wrong order, same work otherwise)
        splat_sort_tile<R>(sh, nullptr, n_tile, lane);
#else
#pragma unroll
        for (int k = 0; k < CH / 64; ++k) sh.near[lane + 64 * k] = (uint8_t)(lane + 64 * k);
        ss_wave_lds_sync();
#endif
        SS_PROF_MARK(3);  // tile sort
#pragma unroll 1
        for (int sb = 0; sb < 8; ++sb) {
            if (!((need >> sb) & 1u)) continue;
            const int sx = (sb >> 2) & 1, sy = (sb >> 1) & 1, sz = sb & 1;
            const bool point_valid = (sx ? pt_ok[0][1] : pt_ok[0][0]) && (sy ? pt_ok[1][1] : pt_ok[1][0]) && (sz ? pt_ok[2][1] : pt_ok[2][0]);
            const R px = sx ? pc[0][1] : pc[0][0], py = sy ? pc[1][1] : pc[1][0], pz = sz ? pc[2][1] : pc[2][0];
            const R slo[3] = {sx ? lo[0][1] : lo[0][0], sy ? lo[1][1] : lo[1][0], sz ? lo[2][1] : lo[2][0]};
            const R shi[3] = {sx ? hi[0][1] : hi[0][0], sy ? hi[1][1] : hi[1][0], sz ? hi[2][1] : hi[2][0]};
            // levelset_grid.fill(0), dense_subdomains.rs:1390, then the sum in index order
            const R acc = splat_accumulate_wave<R, ARITH>(P, sh.pay, sh.wl, n_tile, lane, px, py, pz, slo, shi, P.R2, R(0.0), nullptr, 0, nullptr, sh.near);
            const R val = point_valid ? acc : R(0.0);
            gblock[64 * sb] = val;
            mn = ss_min(mn, val);
            mx = ss_max(mx, val);
            if constexpr (EARLY) faces |= (unsigned long long)splat_face_bits(__ballot(point_valid && !(val > P.threshold))) << (6 * sb);
        }
    }
    SS_PROF_MARK(4);  // exact sums
    int writer = 0;
    if constexpr (sizeof(R) == 4) {
        mn = ss_wave_reduce_to_lane63<false>(mn);
        mx = ss_wave_reduce_to_lane63<true>(mx);
        writer = 63;
    } else {
#pragma unroll
        for (int off2 = 32; off2 > 0; off2 >>= 1) {
            mn = ss_min(mn, __shfl_xor(mn, off2));
            mx = ss_max(mx, __shfl_xor(mx, off2));
        }
    }
    if (lane == writer) {
        blk_minmax[logical] = ss_make2(mn, mx);
        trunc[logical] = EARLY ? certified : (wave_mask == 0xFFu ? 0u : (trunc[logical] & ~wave_mask));
        if constexpr (EARLY) facebits[logical] = faces;
    }
}

// Gather and accumulate in one kernel, ONE WAVE per block: the wave scans the search-cell rows around the block like
// k_splat_gather, keeps the candidates within reach in LDS (payload and particle index, scan order) and goes straight on to
// the classification / ordering / exact sums of splat_accumulate_block_wave -- the tiles of ordinary blocks never pass through
// HBM (S10M-tank: 3.4 GB written and 2.9 GB read back by the two-kernel version) and need no arena.  A block with more than
// SSWaveChunk candidates only reports its count and sets its flag in big[] (the host compacts the flags into a list with a scan): those take the arena path
// (k_splat_bounds, k_splat_gather / _large, k_splat_accumulate_list).  list == nullptr: every active block; otherwise the blocks
// of the device-side list, the sub-blocks in redo_mask only.
#ifndef SS_FUSED_MINWAVES
#define SS_FUSED_MINWAVES 6
#endif
template <class R, int ARITH, bool EARLY>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 4 ? SS_FUSED_MINWAVES : 4, 8))) void k_splat_fused(SSDevT<R> P, const ss_real4<R>* __restrict__ posvol, const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ cell_start, const uint2* __restrict__ row_tab, const uint32_t* __restrict__ active_xyz, uint32_t n_active,
                                                    const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list_dev,
                                                    const uint32_t* __restrict__ redo_mask, R* __restrict__ G, ss_real2<R>* __restrict__ blk_minmax,
                                                    uint32_t* __restrict__ trunc, unsigned long long* __restrict__ facebits, uint32_t* __restrict__ counts,
                                                    uint32_t* __restrict__ big) {
    constexpr int CH = SSWaveChunk<R>::value;
    __shared__ SplatAccWaveShared<R> sh;
    static_assert(sizeof(sh.wl) >= 128 * sizeof(uint32_t), "the row tables of the scan live in the list buffer");
    uint32_t* s_row_start = reinterpret_cast<uint32_t*>(sh.wl);  // (wl is not in use during the scan)
    uint32_t* s_row_prefix = s_row_start + 64;
    const int lane = threadIdx.x;
    const uint32_t n = list ? *n_list_dev : n_active;
    const uint32_t n_slots = ss_xcd_chunked_grid_dev(n);
    auto one_slot = [&](uint32_t w) {
        const uint32_t it = ss_xcd_chunked_group(w);
        if (it >= n) return;
        const uint32_t logical = __builtin_amdgcn_readfirstlane(list ? list[it] : it);
        const int b3[3] = {(int)active_xyz[3 * (size_t)logical], (int)active_xyz[3 * (size_t)logical + 1], (int)active_xyz[3 * (size_t)logical + 2]};
        R plo[3], phi[3];
        uint32_t count = 0, n_near_block = 0;
        SS_PROF_BEGIN();
        ss_wave_lds_sync();  // the previous block's reads of the tile are done
        {
            const uint32_t key0 = splat_block_box<R>(P, b3, plo, phi);
            // Over-dense input: a block whose rows hold more than SS_FUSED_BAIL x the tile capacity is handed to the arena path without
            // scanning (about two thirds of the candidates of a row lie within reach; S10M-cube: the scans of the 89 % of the blocks
            // that overflow anyway cost 1.6 ms).
            uint32_t bailed = 0;
            splat_wave_scan_grouped<R, false>(P, posvol, perm, cell_start, row_tab, key0, plo, phi, s_row_start, s_row_prefix, lane, (uint32_t)(SS_FUSED_BAIL * CH), &bailed,
                                     [&](R d2, uint32_t src, uint32_t, const ss_real4<R>& pv) {
                                         const bool inside = d2 <= P.R2;
                                         const unsigned long long m = __ballot(inside);
                                         const uint32_t pos = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                         if (inside && pos < (uint32_t)CH) {
                                             sh.pay[pos] = pv;
                                             sh.idx[pos] = src;  // (the particle index is looked up by the blocks that order their tile)
                                         }
                                         count += (uint32_t)__popcll(m);
                                         if constexpr (EARLY && sizeof(R) == 4) {
                                             // the certificate works on the entries within the NEAR radius of the block's box only (every sub-block's near
                                             // entries are among them: its box lies inside the block's): their positions in the tile, in sh.near
                                             const bool nb = d2 <= P.R2near;
                                             const unsigned long long mn = __ballot(nb);
                                             if (nb && pos < (uint32_t)CH) sh.near[n_near_block + __builtin_amdgcn_mbcnt_hi((uint32_t)(mn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mn, 0u))] = (uint8_t)pos;
                                             n_near_block += (uint32_t)__popcll(mn);
                                         }
                                         return count <= (uint32_t)CH;  // a block with more candidates takes the arena path, which counts them itself
                                     });
            if (bailed) count = bailed;  // (> CH; the arena path counts exactly)
        }
        if (!list && lane == 0) counts[logical] = count;  // tile entries (statistics; > CH: the arena path recounts)
        if (count > (uint32_t)CH) {
            if (lane == 0) big[logical] = 1u;  // (a flag per block: a list appended to with atomics on one counter serialises -- 132 k appends cost 1.3 ms on S10M-cube)
            return;
        }
        ss_wave_lds_sync();
        SS_PROF_MARK(0);  // candidate scan
#if SS_ABLATE == 1  // (timing / instruction-count knob: the output is wrong)
        if (count < 100000u) return;
#endif
        splat_accumulate_block_wave<R, ARITH, EARLY>(sh, P, logical, (int)count, (int)n_near_block, posvol, perm, active_xyz, G, blk_minmax, trunc, facebits,
                                                           // (the first pass evaluates every sub-block: a compile-time mask removes the second pass's branches from its code)
                                                           EARLY ? 0xFFu : (redo_mask ? (uint32_t)__builtin_amdgcn_readfirstlane(redo_mask[logical]) : 0xFFu));
        SS_PROF_MARK(7);  // whole sub-block walk incl. epilogue (phases 1-6 are inside)
    };
    // first pass: the grid has exactly one workgroup per slot -- written as a loop, the compiler hoists lane constants out of it
    // and spills them to scratch (five stores per block: 1.8 GB per step on S10M-tank)
    if constexpr (EARLY) {
        if (blockIdx.x < n_slots) one_slot(blockIdx.x);
    } else {
        for (uint32_t w = blockIdx.x; w < n_slots; w += gridDim.x) one_slot(w);
    }
}

// the blocks with larger tiles (over-dense input), one workgroup per block; list and its length on the device
template <class R, int ARITH, bool EARLY>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 4 ? 8 : 4, 8))) void k_splat_accumulate_list(SSDevT<R> P, const ss_real4<R>* __restrict__ arena, const uint32_t* __restrict__ arena_idx,
                                                               const unsigned long long* __restrict__ tile_off,
                                                               const uint32_t* __restrict__ counts, const uint32_t* __restrict__ active_xyz,
                                                               const uint32_t* __restrict__ list, const uint32_t* __restrict__ n_list_dev,
                                                               const uint32_t* __restrict__ redo_mask, R* __restrict__ G, ss_real2<R>* __restrict__ blk_minmax,
                                                               uint32_t* __restrict__ trunc, unsigned long long* __restrict__ facebits, bool write_faces, uint32_t* __restrict__ err) {
    __shared__ SplatAccShared<R> sh;
    const uint32_t n = *n_list_dev;
    for (uint32_t it = blockIdx.x; it < n; it += gridDim.x) {
        const uint32_t logical = list[it];
        const uint32_t mask = redo_mask ? redo_mask[logical] : 0xFFu;
        // a block whose tile was not gathered (k_big_tile_select decided nobody would read it) must not be asked for values
        if (err && threadIdx.x == 0 && mask != 0u && tile_off[logical + 1] == tile_off[logical]) atomicOr(err, 4u);  // (a gathered tile has a non-empty reservation)
        splat_accumulate_block<R, ARITH, EARLY>(sh, P, logical, arena, arena_idx, tile_off, counts, active_xyz, G, blk_minmax, trunc, facebits, mask, write_faces);
        __syncthreads();
    }
}

// ---- over-dense blocks: certification straight from the cells -----------------------------------------------------------------
// A block with more candidates than a wave's tile holds (over-dense input: S10M-cube has ~2 200 particles in reach of a block) used
// to get a tile in the arena first and the lower-bound pass then tested all of it against every sub-block.  Most such blocks lie
// deep inside the fluid and need nothing but the certificate, so the certificate now comes first and without a tile: one workgroup
// per block, wave w = sub-block w.  The wave streams the splat cells around ITS sub-block's near box (the box of its 4^3 points
// dilated by the near radius of the classification, not by the particle reach: a quarter of the block's candidates), keeps the
// particles whose box distance is within the near radius as 16-byte records (splat_cert_record) in a list in LDS and puts the list through
// the matrix pipe 32 entries at a time (splat_cert_mfma: the certificate of k_splat_fused); lists longer than SS_CERT_LIST go in pieces, and
// after every piece the wave tests whether all its points are above the threshold already and stops streaming if so.  (Over-dense does
// not mean "certified early": the level set is normalised by the particles' own densities, V = m / rho, so its value inside an over-dense
// fluid is 1 as well and the bound needs the same share of the kernel's mass -- a smaller near radius tried first for dense blocks made
// S10M-cube's certification 5.65 instead of 4.38 ms: it fails and the full radius follows.)
// trunc[b] = the certified sub-blocks; a fully certified block is finished here (min / max, no face bits, nothing stored).
// k_big_tile_select then decides which of these blocks need a tile at all: the ones with a sub-block left to evaluate, and the
// fully certified ones that k_select_redo can ask to complete later -- it only does that for a block with a face neighbour that
// holds (or is) an outside point, i.e. one that is absent or not fully certified itself.
#ifndef SS_CERT_LIST
#define SS_CERT_LIST 192
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_splat_certify_big(SSDevT<float> P, const ss_real4<float>* __restrict__ posvol,
                                                                                                     const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ active_xyz,
                                                                                                     uint32_t n_active, const uint32_t* __restrict__ counts,
                                                                                                     ss_real2<float>* __restrict__ blk_minmax, uint32_t* __restrict__ trunc,
                                                                                                     unsigned long long* __restrict__ facebits, uint32_t* __restrict__ need_mask) {
    __shared__ __attribute__((aligned(16))) uint4 s_list[8][SS_CERT_LIST + 64 + 31];  // certificate records (splat_cert_record); + 31: padding to whole tiles (four workgroups per CU: <= 40 960 bytes)
    __shared__ uint32_t s_row_start[8][64];
    __shared__ uint32_t s_row_prefix[8][64];
    __shared__ uint32_t s_cert;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t logical = ss_xcd_chunked_group(blockIdx.x);
    if (logical >= n_active) return;
    if (counts[logical] <= (uint32_t)SSWaveChunk<float>::value) return;  // k_splat_fused finished this block
    if (tid == 0) s_cert = 0u;
    const int b3[3] = {(int)active_xyz[3 * (size_t)logical], (int)active_xyz[3 * (size_t)logical + 1], (int)active_xyz[3 * (size_t)logical + 2]};
    const int s3[3] = {(wave >> 2) & 1, (wave >> 1) & 1, wave & 1};
    const int o3[3] = {(lane >> 4) & 3, (lane >> 2) & 3, lane & 3};
    float slo[3], shi[3], bc[3];
    bool valid = true, point_valid = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int g0 = b3[d] * SS_BLOCK + 4 * s3[d];
        valid = valid && g0 < P.np[d];
        point_valid = point_valid && (g0 + o3[d]) < P.np[d];
        slo[d] = P.gmin[d] + (float)g0 * P.cs;
        shi[d] = P.gmin[d] + (float)min(g0 + 3, P.np[d] - 1) * P.cs;
        // the records' frame: the centre of THIS WAVE's sub-block (a wave builds its own records: the smaller coordinates cost nothing here and keep the
        // slack of the f16 operands small on coarse grids, where the block's half width is several h -- tools/cert_study.py)
        bc[d] = slo[d] + 1.5f * P.cs;
    }
    // B operands of this wave's two tiles (columns = points [32 g, 32 g + 32) of its sub-block; splat_cert_record): lanes 0-31 hold the x content
    // of their column, lanes 32-63 the (y, z) content.  (The certificate bounds the level set from below whatever arithmetic evaluates it: the
    // points' coordinates are those of the scalar loop, the SIMD loop's z -- one fma -- differs by an ulp, far inside the slack.)
    uint32_t cb0, cb1[2];
    {
        const bool lo_half = lane < 32;
        const float y = ((P.gmin[1] + (float)(b3[1] * SS_BLOCK + 4 * s3[1] + o3[1]) * P.cs) - bc[1]) * P.avx_inv_h;
        const float z = ((P.gmin[2] + (float)(b3[2] * SS_BLOCK + 4 * s3[2] + o3[2]) * P.cs) - bc[2]) * P.avx_inv_h;
        cb0 = ss_pack_f16(lo_half ? 1.0f : y, lo_half ? 1.0f : z);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float x = ((P.gmin[0] + (float)(b3[0] * SS_BLOCK + 4 * s3[0] + 2 * g + ((lane >> 4) & 1)) * P.cs) - bc[0]) * P.avx_inv_h;
            cb1[g] = ss_pack_f16(lo_half ? x : y * y, lo_half ? x * x : z * z);
        }
    }
    bool done = false;
    if (valid) {  // (wave-uniform)
        uint4* list = s_list[wave];
        uint32_t* row_start = s_row_start[wave];
        uint32_t* row_prefix = s_row_prefix[wave];
        const float r2near = P.R2near;
        const float rn = __builtin_amdgcn_sqrtf(r2near) * 1.00001f + P.coord_slack;
        // splat cells overlapping the near box per axis (table-relative)
        int clo[3], chi[3];
        const double cell = 1.0 / P.sinv, cpad = 1.0e-3 * cell;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int a = (int)floor(((double)(slo[d] - rn) - cpad - P.sorg[d]) * P.sinv) - P.kmin[d];
            const int e = (int)floor(((double)(shi[d] + rn) + cpad - P.sorg[d]) * P.sinv) - P.kmin[d];
            clo[d] = max(a, 0);
            chi[d] = min(e, P.kdim[d] - 1);
        }
        const int ny = chi[1] - clo[1] + 1;
        const int nrows = (chi[0] - clo[0] + 1) * ny;
        float acc0 = 0.0f, acc1 = 0.0f;  // this lane's parts of the two tiles' columns (splat_cert_reduce)
        int n_list = 0, n_near = 0;
        // The list so far through the matrix pipe, 32 entries (rows) at a time, both tiles; then the test whether every point of the sub-block is
        // above the threshold ALREADY -- an over-dense fluid is certified by a fraction of its near entries (S10M-cube: ten times the rest density).
        auto flush = [&]() -> bool {
            if (lane < 32) list[n_list + lane] = SS_CERT_DUMMY;
            ss_wave_lds_sync();
            for (int base = 0; base < n_list; base += 32) {
                const uint2 arow = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(list + base + (lane & 31)) + (lane >> 5) * 8);
                const int rows = n_list - base;
                acc0 = splat_cert_reduce(splat_cert_mfma(arow, cb0, cb1[0]), rows, acc0);
                acc1 = splat_cert_reduce(splat_cert_mfma(arow, cb0, cb1[1]), rows, acc1);
            }
            ss_wave_lds_sync();
            n_list = 0;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0), __float_as_uint(acc1), false, false);
            const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            // the margin of thr_inside covers the rounding of the terms; a sum of n of them adds up to n 2^-24 relative
            const float thr = P.thr_inside + ((float)n_near * 1.2e-7f) * P.thr_inside;
            return __ballot(tot > thr || !point_valid) == ~0ull;
        };
        for (int row_base = 0; row_base < nrows; row_base += 64) {
            const int nb = min(64, nrows - row_base);
            uint32_t len = 0;
            if (lane < nb) {
                const int r = row_base + lane;
                const int cx = clo[0] + r / ny, cy = clo[1] + r % ny;
                // distance of the row's (x, y) cell to the sub-block's rectangle; what is left of the near radius along z
                const double x0 = P.sorg[0] + (double)(cx + P.kmin[0]) * cell, y0 = P.sorg[1] + (double)(cy + P.kmin[1]) * cell;
                const double ex = fmax(fmax((double)slo[0] - (x0 + cell), x0 - (double)shi[0]) - cpad, 0.0);
                const double ey = fmax(fmax((double)slo[1] - (y0 + cell), y0 - (double)shi[1]) - cpad, 0.0);
                const double left = (double)rn * (double)rn - (ex * ex + ey * ey);
                uint32_t rb = 0, re = 0;
                if (left >= 0.0) {
                    const double rz = sqrt(left) + cpad;
                    const int zl = max((int)floor(((double)slo[2] - rz - P.sorg[2]) * P.sinv) - P.kmin[2], clo[2]);
                    const int zh = min((int)floor(((double)shi[2] + rz - P.sorg[2]) * P.sinv) - P.kmin[2], chi[2]);
                    if (zl <= zh) {
                        const uint32_t row = (uint32_t)(cx * P.kdim[1] + cy) * (uint32_t)P.kdim[2];
                        rb = cell_start[row + (uint32_t)zl];
                        re = cell_start[row + (uint32_t)zh + 1u];
                    }
                }
                row_start[lane] = rb;
                len = re - rb;
            }
            const uint32_t incl = ss_wave_inclusive_scan(len);
            row_prefix[lane] = (lane < nb) ? incl - len : 0xFFFFFFFFu;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            ss_wave_lds_sync();
            constexpr int GRP = 4;
            for (uint32_t q0 = 0; q0 < total; q0 += 64u * GRP) {
                uint32_t src[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const uint32_t q = min(q0 + 64u * (uint32_t)j + (uint32_t)lane, total - 1u);
                    int lo = 0;  // last row r with row_prefix[r] <= q
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) lo += (row_prefix[lo + step] <= q) ? step : 0;
                    src[j] = row_start[lo] + (q - row_prefix[lo]);
                }
                ss_real4<float> pv[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) pv[j] = posvol[src[j]];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const uint32_t qj = q0 + 64u * (uint32_t)j;
                    if (qj >= total) break;  // (wave-uniform)
                    const float ex = fmaxf(fmaxf(slo[0] - pv[j].x, pv[j].x - shi[0]) - P.coord_slack, 0.0f);
                    const float ey = fmaxf(fmaxf(slo[1] - pv[j].y, pv[j].y - shi[1]) - P.coord_slack, 0.0f);
                    const float ez = fmaxf(fmaxf(slo[2] - pv[j].z, pv[j].z - shi[2]) - P.coord_slack, 0.0f);
                    const bool pass = (qj + (uint32_t)lane < total) && ((ex * ex + ey * ey) + ez * ez <= r2near);
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        if (pass) list[n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = splat_cert_record(P, pv[j], bc[0], bc[1], bc[2], P.cert_e1s, P.cert_e0s);
                        const int c = __popcll(m);
                        n_list += c;
                        n_near += c;
                        if (n_list >= SS_CERT_LIST) done = flush();
                    }
                    if (done) break;  // (wave-uniform)
                }
                if (done) break;
            }
            if (done) break;
            ss_wave_lds_sync();  // the next batch overwrites the row tables
        }
        if (!done && n_list > 0) done = flush();
        if (!done && n_near == 0) done = __ballot(!point_valid) == ~0ull;  // (no entry at all: only a sub-block without valid points is "inside")
    }
    __syncthreads();
    if (done && lane == 0) atomicOr(&s_cert, 1u << wave);
    __syncthreads();
    if (tid == 0) {
        const uint32_t cert = s_cert;
        trunc[logical] = cert;
        if (cert == 0xFFu) {
            blk_minmax[logical] = ss_make2(P.thr_inside, P.thr_inside);
            facebits[logical] = 0ull;
            need_mask[logical] = 0u;
        } else {
            need_mask[logical] = ~cert & 0xFFu;  // (incl. sub-blocks beyond the grid: the exact pass stores their zeros)
        }
    }
}

// Which over-dense blocks need a tile in the arena (see k_splat_certify_big).  (The list of those with sub-blocks left to evaluate is the
// compaction of need_mask != 0, a scan launched by the host.)
// counts[b] = 0 takes a block out of the arena path (k_splat_bounds, k_splat_gather).
template <class R>
__global__ __launch_bounds__(256) void k_big_tile_select(SSDevT<R> P, const uint32_t* __restrict__ active_xyz, uint32_t n_active, const uint32_t* __restrict__ block_slot,
                                                         const uint32_t* __restrict__ trunc, uint32_t* __restrict__ counts) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_active) return;
    if (counts[a] <= (uint32_t)SSWaveChunk<R>::value) return;
    bool needs = trunc[a] != 0xFFu;
    if (!needs) {
        const int b[3] = {(int)active_xyz[3 * (size_t)a], (int)active_xyz[3 * (size_t)a + 1], (int)active_xyz[3 * (size_t)a + 2]};
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                int c[3] = {b[0], b[1], b[2]};
                c[d] += sgn ? 1 : -1;
                if (c[d] < 0 || c[d] >= P.nb[d]) continue;  // beyond the grid: no points there (k_select_redo)
                const uint32_t slot = ss_block_in_table(P, c[0], c[1], c[2]) ? block_slot[ss_block_index(P, c[0], c[1], c[2])] : 0xFFFFFFFFu;
                needs = needs || slot == 0xFFFFFFFFu || trunc[slot] != 0xFFu;
            }
    }
    if (!needs) counts[a] = 0u;
}

// A certified sub-block carries lower bounds, all above the threshold.  Marching cubes classifies with them like with the
// complete values; it needs the value itself only at the end points of edges that cross the surface (dense_subdomains.rs:1516-1517),
// i.e. at points with a 6-neighbour outside.  Such a neighbour of a point of a certified sub-block lies in the face-adjacent
// sub-block, on the touching face: the sub-block is completed iff one of its six neighbours reports an outside point there
// (facebits, written by the first pass).  A neighbour in a block without particles in reach is all zero = outside.
// The statistics of the first pass are taken on the way: stats[0][.] += tile entries, stats[2][.] += certified sub-blocks,
// stats[1][.] += blocks that stay truncated after the second pass (certified sub-blocks nobody reads); 64 copies of every counter.
// trunc == nullptr (no two-pass scheme): only the tile entries are summed.  big[], the flags of the over-dense blocks, are reset for
// the second launch of the splat kernel.
template <class R>
__global__ __launch_bounds__(256) void k_select_redo(SSDevT<R> P, const uint32_t* __restrict__ active_xyz, uint32_t n_active, const uint32_t* __restrict__ block_slot,
                                                     const uint32_t* __restrict__ trunc, const unsigned long long* __restrict__ facebits,
                                                     uint32_t* __restrict__ redo_mask, const uint32_t* __restrict__ counts, unsigned long long* __restrict__ stats,
                                                     uint32_t* __restrict__ big) {
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = a < n_active;
    if (live && big) big[a] = 0u;
    const uint32_t cert = (live && trunc) ? trunc[a] : 0u;
    uint32_t redo = 0;
    if (cert) {
        const int b[3] = {(int)active_xyz[3 * (size_t)a], (int)active_xyz[3 * (size_t)a + 1], (int)active_xyz[3 * (size_t)a + 2]};
        const unsigned long long own = facebits[a];
        // face words of the six neighbour blocks: ~0 for a block without slot, 0 beyond the grid (no points there)
        unsigned long long nbf[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int sgn = 0; sgn < 2; ++sgn) {
                int c[3] = {b[0], b[1], b[2]};
                c[d] += sgn ? 1 : -1;
                unsigned long long w = 0;
                if (c[d] >= 0 && c[d] < P.nb[d]) {  // (a block of the grid outside the table is a block without particles in reach of this rank's points)
                    const uint32_t slot = ss_block_in_table(P, c[0], c[1], c[2]) ? block_slot[ss_block_index(P, c[0], c[1], c[2])] : 0xFFFFFFFFu;
                    w = (slot == 0xFFFFFFFFu) ? ~0ull : facebits[slot];
                }
                nbf[d][sgn] = w;
            }
        for (int sb = 0; sb < 8; ++sb) {
            if (!((cert >> sb) & 1u)) continue;
            bool need = false;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int bit = 4 >> d;  // sub-block coordinate along d: sb = (sx << 2) | (sy << 1) | sz
                const int sd = (sb & bit) ? 1 : 0;
                const int other = sb ^ bit;  // the sub-block on the other side along d (same block or the neighbour block)
                // towards -d: the neighbour's +d face (bit 2d+1); towards +d: its -d face (bit 2d)
                const unsigned long long w_minus = sd ? own : nbf[d][0];
                const unsigned long long w_plus = sd ? nbf[d][1] : own;
                need = need || ((w_minus >> (6 * other + 2 * d + 1)) & 1ull) || ((w_plus >> (6 * other + 2 * d)) & 1ull);
            }
            if (need) redo |= 1u << sb;
        }
    }
    if (live && trunc) redo_mask[a] = redo;
    // statistics
    unsigned long long c0 = live ? (unsigned long long)counts[a] : 0ull, c1 = (cert & ~redo) ? 1ull : 0ull, c2 = (unsigned long long)__popc(cert);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_xor(c0, off);
        c1 += __shfl_xor(c1, off);
        c2 += __shfl_xor(c2, off);
    }
    // one atomic per counter and WORKGROUP, spread over 64 copies of the counters (k_publish_stats adds them up): atomics on one address
    // serialise in its L2 channel -- 65 k of them (one per wave) cost 0.7 ms on S10M-tank
    __shared__ unsigned long long s_part[4][3];
    if ((threadIdx.x & 63u) == 0u) {
        s_part[threadIdx.x >> 6][0] = c0;
        s_part[threadIdx.x >> 6][1] = c1;
        s_part[threadIdx.x >> 6][2] = c2;
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const unsigned long long v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (v) atomicAdd(&stats[(size_t)threadIdx.x * 64u + (blockIdx.x & 63u)], v);
    }
}

// the statistics counters and the lengths of the device-side lists, posted for the host in one dispatch
__global__ void k_publish_stats(const unsigned long long* __restrict__ stats, const uint32_t* __restrict__ n_redo, const uint32_t* __restrict__ n_large, const uint32_t* __restrict__ err,
                                SSMailSlot m0, SSMailSlot m1, SSMailSlot m2, SSMailSlot m3) {
    unsigned long long t[3] = {0ull, 0ull, 0ull};
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 64; ++j) t[k] += stats[k * 64 + j];
    ss_mail_post(m0, t[0]);
    ss_mail_post(m1, t[1]);
    ss_mail_post(m2, t[2]);
    const unsigned long long r = (unsigned long long)(n_redo[0] & 0xFFFFFFFu), l = (unsigned long long)(n_large[0] & 0xFFFFFFFu), e = (unsigned long long)(err[0] & 0xFFu);
    ss_mail_post(m3, r | (l << 28) | (e << 56));
}
void ss_launch_publish_stats(const unsigned long long* stats, const uint32_t* n_redo, const uint32_t* n_large, const uint32_t* err, SSMailSlot m0, SSMailSlot m1, SSMailSlot m2, SSMailSlot m3,
                             hipStream_t st) {
    hipLaunchKernelGGL(k_publish_stats, dim3(1), dim3(1), 0, st, stats, n_redo, n_large, err, m0, m1, m2, m3);
}

template <class R>
void ss_launch_splat_bounds(const SSDevT<R>& P, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* counts, uint32_t* bound, hipStream_t st) {
    hipLaunchKernelGGL(k_splat_bounds<R>, dim3((n_active + 1u + 255u) / 256u), dim3(256), 0, st, P, cell_start, active_xyz, n_active, counts, bound);
}

template <class R>
void ss_launch_splat_gather(const SSDevT<R>& P, const ss_real4<R>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active,
                            const unsigned long long* tile_off, ss_real4<R>* arena, uint32_t* arena_idx, uint32_t* counts, uint32_t* large_flag, hipStream_t st) {
    if (!n_active) return;
    const uint32_t n_groups = (n_active + 3u) / 4u;
    hipLaunchKernelGGL(k_splat_gather<R>, dim3(ss_xcd_chunked_grid(n_groups)), dim3(256), 0, st, P, posvol, perm, cell_start, active_xyz, n_active, tile_off, arena, arena_idx, counts,
                       large_flag);
}

// the blocks k_splat_gather flagged (list and its length on the device)
template <class R>
void ss_launch_splat_gather_large(const SSDevT<R>& P, const ss_real4<R>* posvol, const ss_real4<R>* posvol_by_index, const uint32_t* perm, const uint32_t* cell_start,
                                  const uint32_t* active_xyz, const uint32_t* large_list, const uint32_t* n_large_dev, const uint32_t* counts,
                                  const unsigned long long* tile_off, ss_real4<R>* arena, uint32_t* arena_idx, hipStream_t st) {
    hipLaunchKernelGGL((k_splat_gather_large<R, SSTileCap<R>::value>), dim3(2048), dim3(512), 0, st, P, posvol, posvol_by_index, perm, cell_start, active_xyz, large_list,
                       n_large_dev, counts, tile_off, arena, arena_idx);
}

#define SS_SPLAT_DISPATCH(LAUNCH)                                  \
    do {                                                           \
        if constexpr (sizeof(R) == 4) {                            \
            switch (P.arith) {                                     \
                case SS_ARITH_FAST: LAUNCH(SS_ARITH_FAST); return; \
                case SS_ARITH_SIMD: LAUNCH(SS_ARITH_SIMD); return; \
                case SS_ARITH_SIMD_LEAN: LAUNCH(SS_ARITH_SIMD_LEAN); return; \
                case SS_ARITH_SIMD_HW: LAUNCH(SS_ARITH_SIMD_HW); return;     \
                default: break;                                    \
            }                                                      \
        }                                                          \
        LAUNCH(SS_ARITH_GENERIC);                                  \
    } while (0)

// The splat of the ordinary blocks (k_splat_fused).  list == nullptr: first pass over all n_active blocks (lower-bound
// certification unless full_levelset); otherwise the second pass over the device-side list of blocks with certified sub-blocks that
// marching cubes reads.  `big`: n_active flags, set for the blocks with more than SSWaveChunk candidates (compacted by the host into the
// list ss_launch_splat_accumulate_big works on); counts: tile size per block (first pass).
template <class R>
void ss_launch_splat_fused(const SSDevT<R>& P, const ss_real4<R>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint2* row_tab, const uint32_t* active_xyz, uint32_t n_active,
                           R* G, ss_real2<R>* blk_minmax, uint32_t* trunc, bool full_levelset, const uint32_t* list, const uint32_t* n_list_dev,
                           const uint32_t* redo_mask, unsigned long long* facebits, uint32_t* counts, uint32_t* big, hipStream_t st) {
    if (!n_active) return;
    // (big[] must be 0: the host takes the flags from its zeroed words for the first launch, k_select_redo resets them for the second)
    const dim3 grid(list ? 32768u : ss_xcd_chunked_grid(n_active));
#define SS_FUSED(A)                                                                                                                                          \
    do {                                                                                                                                                     \
        if (list || full_levelset)                                                                                                                           \
            hipLaunchKernelGGL((k_splat_fused<R, A, false>), grid, dim3(64), 0, st, P, posvol, perm, cell_start, row_tab, active_xyz, n_active, list, n_list_dev,    \
                               redo_mask, G, blk_minmax, trunc, facebits, counts, big);                                                                      \
        else                                                                                                                                                 \
            hipLaunchKernelGGL((k_splat_fused<R, A, true>), grid, dim3(64), 0, st, P, posvol, perm, cell_start, row_tab, active_xyz, n_active, list, n_list_dev,     \
                               redo_mask, G, blk_minmax, trunc, facebits, counts, big);                                                                      \
    } while (0)
    SS_SPLAT_DISPATCH(SS_FUSED);
#undef SS_FUSED
}

// The blocks k_splat_fused handed on (big[0] of them, big[1..]): their tiles are in the arena (k_splat_gather / _large), one
// workgroup per block.  second_pass: only the sub-blocks in redo_mask.  exact_first: the first pass after k_splat_certify_big --
// `big` is the list of blocks with sub-blocks left to evaluate, redo_mask those sub-blocks; exact sums only, face bits written.
template <class R>
void ss_launch_splat_accumulate_big(const SSDevT<R>& P, const ss_real4<R>* arena, const uint32_t* arena_idx, const unsigned long long* tile_off, const uint32_t* counts,
                                    const uint32_t* active_xyz, R* G, ss_real2<R>* blk_minmax, uint32_t* trunc, bool full_levelset, bool second_pass, bool exact_first,
                                    const uint32_t* redo_mask, unsigned long long* facebits, const uint32_t* big, uint32_t* err, hipStream_t st) {
    const dim3 lgrid(2048);
#define SS_BIG(A)                                                                                                                                              \
    do {                                                                                                                                                       \
        if (second_pass || full_levelset || exact_first)                                                                                                       \
            hipLaunchKernelGGL((k_splat_accumulate_list<R, A, false>), lgrid, dim3(512), 0, st, P, arena, arena_idx, tile_off, counts, active_xyz, big + 1, big, \
                               (second_pass || exact_first) ? redo_mask : nullptr, G, blk_minmax, trunc, facebits, exact_first, err);                         \
        else                                                                                                                                                   \
            hipLaunchKernelGGL((k_splat_accumulate_list<R, A, true>), lgrid, dim3(512), 0, st, P, arena, arena_idx, tile_off, counts, active_xyz, big + 1, big, \
                               nullptr, G, blk_minmax, trunc, facebits, false, err);                                                                          \
    } while (0)
    SS_SPLAT_DISPATCH(SS_BIG);
#undef SS_BIG
}

// Over-dense blocks of an f32 job with the two-pass scheme: certificates first (k_splat_certify_big), then the choice of the blocks that
// get a tile (k_big_tile_select).  need_mask must be zero on entry (it is read for every active block afterwards).
void ss_launch_splat_certify_big(const SSDevT<float>& P, const ss_real4<float>* posvol, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active,
                                 const uint32_t* block_slot, uint32_t* counts, ss_real2<float>* blk_minmax, uint32_t* trunc, unsigned long long* facebits, uint32_t* need_mask,
                                 hipStream_t st) {
    if (!n_active) return;
    hipLaunchKernelGGL(k_splat_certify_big, dim3(ss_xcd_chunked_grid(n_active)), dim3(512), 0, st, P, posvol, cell_start, active_xyz, n_active, counts, blk_minmax, trunc, facebits,
                       need_mask);
    hipLaunchKernelGGL(k_big_tile_select<float>, dim3((n_active + 255u) / 256u), dim3(256), 0, st, P, active_xyz, n_active, block_slot, trunc, counts);
}

template <class R>
void ss_launch_select_redo(const SSDevT<R>& P, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* block_slot, const uint32_t* trunc,
                           const unsigned long long* facebits, uint32_t* redo_mask, const uint32_t* counts, unsigned long long* stats, uint32_t* big, hipStream_t st) {
    hipLaunchKernelGGL(k_select_redo<R>, dim3((n_active + 255u) / 256u), dim3(256), 0, st, P, active_xyz, n_active, block_slot, trunc, facebits, redo_mask, counts, stats, big);
}

// =====================================================================================================
// K4/K5: marching cubes.  One 512-thread workgroup per MC block b: thread t <-> grid point
// (t>>6, (t>>3)&7, t&7) of block b, owning the cell with that origin and the three edges leaving
// it in +x/+y/+z.  The 9^3 points needed come from the level-set blocks b+{0,1}^3 (absent => 0).
// A block's vertices are numbered axis-major; the per-axis crossing masks (one 64-bit ballot per wave)
// are stored so that neighbouring blocks can compute vertex ids of shared edges without a hash map:
//   id(point p, axis a) = vbase[block] + #crossings of axes < a + popcount(mask_a below p).
// =====================================================================================================
template <class R>
struct McTile {
    R g[9 * 9 * 9];
};

// The eight blocks m + {0,1}^3 of every MC block, looked up once (SS_MC_REC words per MC block): mc_nb[n] = level-set slot of
// neighbour n = (dx << 2) | (dy << 1) | dz (0xFFFFFFFF: no such block), mc_nb[8 + n] = mask of its 4^3 sub-blocks that the splat
// certified to lie inside the surface and never evaluated in full, mc_nb[16 + n] = its slot in the MC list (crossing masks,
// vertex base; 0xFFFFFFFF: not triangulated).  The count and the emit kernel then reach the level-set values with two
// dependent loads (record, value) instead of four (block coordinates, slot, mask, value) -- they are latency-bound.
// mc_nb[24 + d] = subdomain of the block's first point along axis d (the emit kernel's per-vertex divisions start from it; the
// division itself, wave-uniform but done by the vector unit, cost every wave of that kernel 75 instructions).  SS_MC_REC: ss_kernels.h.
template <class R>
__global__ __launch_bounds__(256) void k_mc_neighbours(SSDevT<R> P, const uint32_t* __restrict__ mc_xyz, uint32_t n_mc, const uint32_t* __restrict__ block_slot,
                                                       const uint32_t* __restrict__ mc_slot, const uint32_t* __restrict__ certified, uint32_t* __restrict__ mc_nb) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = t >> 3, n = t & 7u;
    if (m >= n_mc) return;
    const int x = (int)mc_xyz[3 * (size_t)m] + (int)((n >> 2) & 1u), y = (int)mc_xyz[3 * (size_t)m + 1] + (int)((n >> 1) & 1u), z = (int)mc_xyz[3 * (size_t)m + 2] + (int)(n & 1u);
    uint32_t slot = 0xFFFFFFFFu, cert = 0u, mslot = 0xFFFFFFFFu;
    if (ss_block_in_table(P, x, y, z)) {
        const size_t b = ss_block_index(P, x, y, z);
        slot = block_slot[b];
        mslot = mc_slot[b];
        if (slot != 0xFFFFFFFFu && certified) cert = certified[slot];
    }
    mc_nb[SS_MC_REC * (size_t)m + n] = slot;
    mc_nb[SS_MC_REC * (size_t)m + 8 + n] = cert;
    mc_nb[SS_MC_REC * (size_t)m + 16 + n] = mslot;
    if (n < 4u) mc_nb[SS_MC_REC * (size_t)m + 24 + n] = (n < 3u) ? (uint32_t)(((int)mc_xyz[3 * (size_t)m + n] * SS_BLOCK) / P.n_sub_cubes) : 0u;
}

// s_nb: the block's record of mc_nb in LDS.  Points of a certified sub-block read as "a value above the threshold" (they are no
// end point of an edge that crosses the surface, k_select_redo, so the value itself is never used); absent blocks are all zero.
template <class R>
__device__ __forceinline__ R mc_fetch_point(const SSDevT<R>& P, const R* __restrict__ G, const uint32_t* s_nb, int x, int y, int z) {
    const int n = ((x >> 3) << 2) | ((y >> 3) << 1) | (z >> 3);
    const uint32_t slot = s_nb[n];
    if (slot == 0xFFFFFFFFu) return R(0.0);
    const int sbit = (((x & 7) >> 2) << 2) | (((y & 7) >> 2) << 1) | ((z & 7) >> 2);
    if ((s_nb[8 + n] >> sbit) & 1u) return P.thr_inside;
    return G[(size_t)slot * SS_BLOCK_POINTS + (size_t)SS_BLOCK_OFFSET(x & 7, y & 7, z & 7)];
}
// point (x, y, z) of the block that is stored at offset `off` (inverse of SS_BLOCK_OFFSET)
__device__ __forceinline__ void mc_point_of_offset(int off, int* x, int* y, int* z) {
    const int sb = off >> 6;
    *x = ((sb >> 2) & 1) * 4 + ((off >> 4) & 3);
    *y = ((sb >> 1) & 1) * 4 + ((off >> 2) & 3);
    *z = (sb & 1) * 4 + (off & 3);
}

// 729 points for 512 threads.  Thread t reads the value at offset t of the block itself -- one contiguous 2-KiB read per workgroup
// instead of 16-byte runs in (x, y, z) order -- and threads 0..216 one point of the halo (x = 8, then y = 8, then z = 8); both
// loads are issued before either value is written to LDS.
template <class R>
__device__ inline void mc_load_tile(McTile<R>& t, const SSDevT<R>& P, const R* __restrict__ G, const uint32_t* s_nb, int tid) {
    int x0, y0, z0;
    mc_point_of_offset(tid, &x0, &y0, &z0);
    const R v0 = mc_fetch_point(P, G, s_nb, x0, y0, z0);
    int x1 = 8, y1 = 0, z1 = 0;
    if (tid < 81) {  // plane x = 8
        y1 = tid / 9;
        z1 = tid % 9;
    } else if (tid < 153) {  // plane y = 8, x < 8
        x1 = (tid - 81) / 9;
        y1 = 8;
        z1 = (tid - 81) % 9;
    } else {  // plane z = 8, x < 8, y < 8
        x1 = ((tid - 153) >> 3) & 7;
        y1 = (tid - 153) & 7;
        z1 = 8;
    }
    R v1 = R(0.0);
    if (tid < 217) v1 = mc_fetch_point(P, G, s_nb, x1, y1, z1);
    t.g[(x0 * 9 + y0) * 9 + z0] = v0;
    if (tid < 217) t.g[(x1 * 9 + y1) * 9 + z1] = v1;
}

// per-thread classification shared by the count and emit kernels
struct McLocal {
    int gx, gy, gz;       // global point index of this thread
    bool cross[3];        // edge leaving the point along axis a crosses the iso-surface
    int case_index;       // 0 if the cell does not exist
    int ntri;
};

// ntri_of_case: reads the caller's LDS copy of the packed case table (c_mc_packed) -- read from constant memory, the row of a cell's case is a third
// dependent round trip at the end of a kernel that is bound by its first two
template <class R, class NT>
__device__ inline McLocal mc_classify(const McTile<R>& t, const SSDevT<R>& P, int bx, int by, int bz, int tid, NT ntri_of_case) {
    McLocal L;
    const int lx = tid >> 6, ly = (tid >> 3) & 7, lz = tid & 7;
    L.gx = bx * SS_BLOCK + lx;
    L.gy = by * SS_BLOCK + ly;
    L.gz = bz * SS_BLOCK + lz;
    // points / edges / cells of the shard region only (full domain: pt_hi = np - 1)
    const bool point_exists = L.gx <= P.pt_hi[0] && L.gy <= P.pt_hi[1] && L.gz <= P.pt_hi[2];
    const R thr = P.threshold;
    // the eight points of the cell, read once at constant offsets from the thread's own point (all 9^3 entries of the tile are
    // filled, mc_load_tile*): v[dx][dy][dz]
    const R* p = &t.g[(lx * 9 + ly) * 9 + lz];
    const bool b000 = p[0] > thr, b001 = p[1] > thr, b010 = p[9] > thr, b011 = p[10] > thr;  // dense_subdomains.rs:1482 (strict >)
    const bool b100 = p[81] > thr, b101 = p[82] > thr, b110 = p[90] > thr, b111 = p[91] > thr;
    L.cross[0] = point_exists && (L.gx + 1 <= P.pt_hi[0]) && (b000 != b100);
    L.cross[1] = point_exists && (L.gy + 1 <= P.pt_hi[1]) && (b000 != b010);
    L.cross[2] = point_exists && (L.gz + 1 <= P.pt_hi[2]) && (b000 != b001);
    const bool cell_exists = L.gx < P.pt_hi[0] && L.gy < P.pt_hi[1] && L.gz < P.pt_hi[2];
    // corner c of the cell is the point c_corner[c] (uniform_grid.rs:825-834), bit c of the case index says "above the threshold"
    // (marching_cubes_lut.rs:322-329): corners 0..7 = 000, 100, 110, 010, 001, 101, 111, 011 in (dx, dy, dz)
    const int bits = (b000 ? 1 : 0) | (b100 ? 2 : 0) | (b110 ? 4 : 0) | (b010 ? 8 : 0) | (b001 ? 16 : 0) | (b101 ? 32 : 0) | (b111 ? 64 : 0) | (b011 ? 128 : 0);
    L.case_index = cell_exists ? bits : 0;  // 0 if the cell does not exist
    L.ntri = ntri_of_case(L.case_index);  // triangles of the case, from the caller's copy of the packed table in LDS
    return L;
}

// the x-slabs 4 half .. 4 half + 4 of the 9^3 tile (what the points with x in 4 half .. 4 half + 3 and their cells read), 256 threads:
// thread t reads offset 256 half + t of the block itself (the four sub-blocks with that x half are contiguous) and threads
// 0..148 one of the other points: the slab x = 4 half + 4 (y, z < 8), then y = 8, then z = 8 (y < 8)
template <class R>
__device__ inline void mc_load_half_tile(McTile<R>& t, const SSDevT<R>& P, const R* __restrict__ G, const uint32_t* s_nb, int half, int tid) {
    int x0, y0, z0;
    mc_point_of_offset(256 * half + tid, &x0, &y0, &z0);
    const R v0 = mc_fetch_point(P, G, s_nb, x0, y0, z0);
    int x1, y1, z1;
    if (tid < 64) {
        x1 = 4 * half + 4;
        y1 = tid >> 3;
        z1 = tid & 7;
    } else if (tid < 109) {
        x1 = 4 * half + (tid - 64) / 9;
        y1 = 8;
        z1 = (tid - 64) % 9;
    } else {
        x1 = 4 * half + (tid - 109) / 8;
        y1 = (tid - 109) % 8;
        z1 = 8;
    }
    R v1 = R(0.0);
    if (tid < 149) v1 = mc_fetch_point(P, G, s_nb, x1, y1, z1);
    t.g[(x0 * 9 + y0) * 9 + z0] = v0;
    if (tid < 149) t.g[(x1 * 9 + y1) * 9 + z1] = v1;
}

// all 729 points for 256 threads: thread t reads offsets t and 256 + t of the block itself (two contiguous 1-KiB reads per workgroup)
// and threads 0..216 one point of the halo; the three loads are issued before any value is written to LDS
template <class R>
__device__ inline void mc_load_tile_256(McTile<R>& t, const SSDevT<R>& P, const R* __restrict__ G, const uint32_t* s_nb, int tid) {
    int xa, ya, za, xb, yb, zb;
    mc_point_of_offset(tid, &xa, &ya, &za);
    mc_point_of_offset(256 + tid, &xb, &yb, &zb);
    const R va = mc_fetch_point(P, G, s_nb, xa, ya, za);
    const R vb = mc_fetch_point(P, G, s_nb, xb, yb, zb);
    int x1 = 8, y1 = 0, z1 = 0;
    if (tid < 81) {  // plane x = 8
        y1 = tid / 9;
        z1 = tid % 9;
    } else if (tid < 153) {  // plane y = 8, x < 8
        x1 = (tid - 81) / 9;
        y1 = 8;
        z1 = (tid - 81) % 9;
    } else {  // plane z = 8, x < 8, y < 8
        x1 = ((tid - 153) >> 3) & 7;
        y1 = (tid - 153) & 7;
        z1 = 8;
    }
    R v1 = R(0.0);
    if (tid < 217) v1 = mc_fetch_point(P, G, s_nb, x1, y1, z1);
    t.g[(xa * 9 + ya) * 9 + za] = va;
    t.g[(xb * 9 + yb) * 9 + zb] = vb;
    if (tid < 217) t.g[(x1 * 9 + y1) * 9 + z1] = v1;
}

// all 729 points for 128 threads: thread t reads offsets t, 128 + t, 256 + t, 384 + t of the block itself (four contiguous 512-byte
// reads per workgroup) and up to two points of the halo; all loads are issued before any value is written to LDS
template <class R>
__device__ inline void mc_load_tile_128(McTile<R>& t, const SSDevT<R>& P, const R* __restrict__ G, const uint32_t* s_nb, int tid) {
    R v[4];
    int px[4], py[4], pz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mc_point_of_offset(128 * q + tid, &px[q], &py[q], &pz[q]);
        v[q] = mc_fetch_point(P, G, s_nb, px[q], py[q], pz[q]);
    }
    R h[2] = {R(0.0), R(0.0)};
    int hx[2], hy[2], hz[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int k = 128 * q + tid;
        hx[q] = 8, hy[q] = 0, hz[q] = 0;
        if (k < 81) {  // plane x = 8
            hy[q] = k / 9;
            hz[q] = k % 9;
        } else if (k < 153) {  // plane y = 8, x < 8
            hx[q] = (k - 81) / 9;
            hy[q] = 8;
            hz[q] = (k - 81) % 9;
        } else {  // plane z = 8, x < 8, y < 8
            hx[q] = ((k - 153) >> 3) & 7;
            hy[q] = (k - 153) & 7;
            hz[q] = 8;
        }
        if (k < 217) h[q] = mc_fetch_point(P, G, s_nb, hx[q], hy[q], hz[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) t.g[(px[q] * 9 + py[q]) * 9 + pz[q]] = v[q];
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (128 * q + tid < 217) t.g[(hx[q] * 9 + hy[q]) * 9 + hz[q]] = h[q];
}

// ONE 128-thread workgroup per MC block, four points per thread (wave w: the x-slabs w, 2 + w, 4 + w, 6 + w).  The kernel is bound by the
// latency of its two dependent round trips (record, values): a block's round trips are paid once (round 2: two workgroups of 256
// per block paid them and the record twice), sixteen such workgroups fit a CU.  Counts are plain stores.
template <class R>
__global__ __launch_bounds__(128) void k_mc_count(SSDevT<R> P, const R* __restrict__ G, const uint32_t* __restrict__ mc_nb,
                                                  const uint32_t* __restrict__ mc_xyz, uint32_t n_mc, unsigned long long* __restrict__ masks,
                                                  uint32_t* __restrict__ vcount, uint32_t* __restrict__ tcount) {
    __shared__ McTile<R> tile;
    __shared__ uint32_t s_nb[SS_MC_REC];
    __shared__ uint32_t s_v[2], s_t[2];
    __shared__ uint32_t s_ntri4[32];  // triangles per case, a nibble each (mc_classify)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t m = blockIdx.x;
    if (m >= n_mc) return;
    const int bx = (int)mc_xyz[3 * (size_t)m], by = (int)mc_xyz[3 * (size_t)m + 1], bz = (int)mc_xyz[3 * (size_t)m + 2];  // (block coordinates: SSBlockListOut)
    if (tid >= 96) s_ntri4[tid - 96] = c_mc_packed.ntri4[tid - 96];
    if (tid < SS_MC_REC) s_nb[tid] = mc_nb[SS_MC_REC * (size_t)m + tid];
    __syncthreads();
    mc_load_tile_128(tile, P, G, s_nb, tid);
    __syncthreads();
    uint32_t nv = 0, ntri = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const McLocal L = mc_classify(tile, P, bx, by, bz, 128 * q + tid, [&](int ci) { return (int)((s_ntri4[ci >> 3] >> (4 * (ci & 7))) & 7u); });
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned long long mk = __ballot(L.cross[a]);
            if (lane == 0) masks[(size_t)m * 24 + a * 8 + 2 * q + wave] = mk;
            nv += (uint32_t)__popcll(mk);
        }
        ntri += (uint32_t)L.ntri;
    }
    // triangles of this wave (lane 63 of the DPP scan holds the sum)
    const uint32_t nt = ss_wave_inclusive_scan(ntri);
    if (lane == 63) {
        s_v[wave] = nv;
        s_t[wave] = nt;
    }
    __syncthreads();
    if (tid == 0) {
        vcount[m] = s_v[0] + s_v[1];
        tcount[m] = s_t[0] + s_t[1];
    }
}

// ONE 256-thread workgroup per MC block, two points per thread like k_mc_count (wave w: the x-slabs w and 4 + w; eight such
// workgroups per CU instead of four of 512 threads, the two round trips of a block -- record, then values and masks -- paid once).
// Vertices and triangles are both emitted from RECORDS in LDS: a surface block has ~35 vertices and ~70 triangles on its 512 points, so
// "every lane handles the crossings of its own point" ran the vertex arithmetic (a division, three coordinates) 24 times per block for
// a few lanes each, with scattered 12-byte stores.  Instead every crossing files (point, axis) at its rank within the block -- the
// rank is the vertex id minus the block's base -- and lane k then builds vertex k: one or two trips per block, contiguous stores.
template <class R>
__global__ __launch_bounds__(256) void k_mc_emit(SSDevT<R> P, const R* __restrict__ G, const uint32_t* __restrict__ mc_nb,
                                                 const uint32_t* __restrict__ mc_xyz, const uint32_t* __restrict__ mc_slot, uint32_t n_mc,
                                                 const unsigned long long* __restrict__ masks, const uint32_t* __restrict__ vbase,
                                                 const uint32_t* __restrict__ tbase, R* __restrict__ vertices,
                                                 unsigned long long* __restrict__ vkeys, uint32_t* __restrict__ triangles) {
    __shared__ McTile<R> tile;
    __shared__ uint32_t s_nb[SS_MC_REC];
    __shared__ unsigned long long s_mask[8][24];  // [neighbour][axis*8+word]
    __shared__ uint32_t s_pref[8][24];            // vertices of that neighbour block before (axis, word)
    __shared__ uint32_t s_vbase[8];
    __shared__ uint32_t s_tslab[8];        // triangles of x-slab s
    __shared__ uint32_t s_rec[8][5 * 64];  // per x-slab: (cell, triangle number, case) of its triangles, in cell order
    __shared__ unsigned long long s_row[256];  // the packed case table: the look-ups of a trip hit 64 different rows
    uint16_t* s_vrec = reinterpret_cast<uint16_t*>(&s_rec[0][0]);  // (point, axis) of the block's vertices by rank, 3 x 512 at most: in the triangle records' space, before those are filed
    static_assert(sizeof(s_rec) >= 3 * 512 * sizeof(uint16_t), "vertex records share the triangle records' space");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t m = blockIdx.x;
    if (m >= n_mc) return;
    // everything the first round trip can fetch is requested before the "nothing to emit" test waits for its four words
    const uint32_t vb0 = vbase[m], vb1 = vbase[m + 1], tb0 = tbase[m], tb1 = tbase[m + 1];
    const unsigned long long lut_word = c_mc_packed.row[tid];
    const int bx = (int)mc_xyz[3 * (size_t)m], by = (int)mc_xyz[3 * (size_t)m + 1], bz = (int)mc_xyz[3 * (size_t)m + 2];  // (block coordinates: SSBlockListOut)
    uint32_t nb_word = 0;
    if (tid < SS_MC_REC) nb_word = mc_nb[SS_MC_REC * (size_t)m + tid];
    if (vb1 == vb0 && tb1 == tb0) return;  // nothing to emit for this block
    s_row[tid] = lut_word;
    if (tid < SS_MC_REC) s_nb[tid] = nb_word;
    __syncthreads();
    mc_load_tile_256(tile, P, G, s_nb, tid);
    // crossing masks of this block and its 7 upper neighbours, and for each (neighbour, word) the neighbour's vertices before that
    // word: thread 32 nb + w holds word w of neighbour nb, and the prefix is a DPP scan over the wave's two neighbours (eight
    // threads walking 24 words one after the other held the others up for 24 LDS round trips)
    {
        const int nb = tid >> 5, w = tid & 31;
        unsigned long long mk = 0;
        uint32_t slot = 0xFFFFFFFFu;
        if (w < 24) {
            slot = s_nb[16 + nb];
            if (slot != 0xFFFFFFFFu) mk = masks[(size_t)slot * 24 + w];
        }
        const uint32_t c = (uint32_t)__popcll(mk);
        uint32_t incl = ss_wave_inclusive_scan(c);
        const uint32_t first_half = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
        if (lane >= 32) incl -= first_half;
        if (w < 24) {
            s_mask[nb][w] = mk;
            s_pref[nb][w] = incl - c;
            if (w == 0) s_vbase[nb] = (slot != 0xFFFFFFFFu) ? vbase[slot] : 0u;
        }
    }
    __syncthreads();
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // ---- classification; the crossings of the three edges a point owns filed by rank (dense_subdomains.rs:1498-1539) ----
    int case_of[2], ntri_of[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int pt = 256 * half + tid, slab = 4 * half + wave;
        const McLocal L = mc_classify(tile, P, bx, by, bz, pt, [&](int ci) { return (int)(s_row[ci] >> 60); });
        case_of[half] = L.case_index;
        ntri_of[half] = L.ntri;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (L.cross[a]) s_vrec[s_pref[0][a * 8 + slab] + (uint32_t)__popcll(s_mask[0][a * 8 + slab] & below)] = (uint16_t)(pt | (a << 9));
    }
    __syncthreads();
    // ---- vertices: lane k builds the block's k-th vertex ----
    {
        const int n = P.n_sub_cubes;
        // subdomain of the block's first point per axis (k_mc_neighbours): the points of the block (and the point before the first one)
        // are at most one subdomain border away from it when a subdomain has more than 8 cubes, so the per-vertex divisions below
        // (~25 instructions each) become a comparison
        const int g0[3] = {bx * SS_BLOCK, by * SS_BLOCK, bz * SS_BLOCK};
        int q0[3], r0[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            q0[d] = (int)s_nb[24 + d];
            r0[d] = g0[d] - q0[d] * n;
        }
        const bool near_border_rule = n > SS_BLOCK;  // (wave-uniform)
        auto subdomain_of = [&](int d, int x) {  // x / n for a point index x in [g0[d] - 1, g0[d] + 8], x >= 0
            if (!near_border_rule) return x / n;
            const int t = r0[d] + (x - g0[d]);
            return q0[d] + (t >= n ? 1 : 0) - (t < 0 ? 1 : 0);
        };
        // global edge key = 3 x (flat index of the origin point) + axis: block-uniform base plus small multiples of the strides
        const unsigned long long stride_y = 3ull * (unsigned long long)P.np[2], stride_x = stride_y * (unsigned long long)P.np[1];
        const unsigned long long block_key = ((unsigned long long)g0[0] * (unsigned long long)P.np[1] + (unsigned long long)g0[1]) * stride_y + 3ull * (unsigned long long)g0[2];
        const uint32_t nv = vb1 - vb0;  // (= the crossings of the block's 24 mask words, k_mc_count)
        for (uint32_t k = (uint32_t)tid; k < nv; k += 256u) {
            const uint32_t rec = s_vrec[k];
            const int pt = (int)(rec & 511u), a = (int)(rec >> 9);
            const int l3[3] = {pt >> 6, (pt >> 3) & 7, pt & 7};
            const int at = (l3[0] * 9 + l3[1]) * 9 + l3[2];
            const R ov = tile.g[at];
            const R tv = tile.g[at + (a == 0 ? 81 : (a == 1 ? 9 : 1))];
            const R alpha = (P.threshold - ov) / (tv - ov);  // :1516-1517
            R vc[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // coordinates in the marching-cubes grid of the lowest-index subdomain that generates this
                // vertex ("first patch wins" with patches in ascending flat subdomain index, :1707-1716):
                // the subdomain of the adjacent cell O - delta, delta_d = 1 on the orthogonal axes where possible
                const int O = g0[d] + l3[d];
                const int xq = O - (d == a ? 0 : 1);
                const int sd = (xq >= 0) ? subdomain_of(d, xq) : 0;
                const int loc = O - sd * n;
                const R sub_min = P.gmin[d] + (R)sd * P.sub_size;  // uniform_grid.rs:454-467 on the subdomain grid
                const R oc = sub_min + (R)loc * P.cs;               // uniform_grid.rs:418-425 on the subdomain MC grid
                const R tc = sub_min + (R)(loc + (d == a ? 1 : 0)) * P.cs;
                vc[d] = oc * (R(1.0) - alpha) + tc * alpha;  // :1518-1519
            }
            const size_t vid = (size_t)vb0 + k;
            vertices[3 * vid] = vc[0];
            vertices[3 * vid + 1] = vc[1];
            vertices[3 * vid + 2] = vc[2];
            vkeys[vid] = block_key + (unsigned long long)(uint32_t)l3[0] * stride_x + (unsigned long long)(uint32_t)l3[1] * stride_y + (unsigned long long)(3 * l3[2] + a);
        }
    }
    __syncthreads();  // the vertex records are read: their space takes the triangle records
    // ---- triangle records of the slabs ----
    // Only one cell in eight of a surface block has triangles: a loop "for my cell's triangles" keeps a few lanes busy for five
    // trips and scatters 12-byte stores.  Instead every lane files a record (cell, case, triangle number) per triangle of its
    // cell at the triangle's rank within the slab, and the wave then emits records 64 at a time: lane k builds the k-th
    // triangle, so the stores of a trip form one contiguous run.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int pt = 256 * half + tid, slab = 4 * half + wave;
        const uint32_t incl = ss_wave_inclusive_scan((uint32_t)ntri_of[half]);  // (DPP: no LDS round trips)
        if (lane == 63) s_tslab[slab] = incl;
        const uint32_t excl = incl - (uint32_t)ntri_of[half];
        for (int i = 0; i < ntri_of[half]; ++i) s_rec[slab][excl + (uint32_t)i] = (uint32_t)pt | ((uint32_t)i << 9) | ((uint32_t)case_of[half] << 12);
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int slab = 4 * half + wave;
        uint32_t toff = tb0;
        for (int w = 0; w < slab; ++w) toff += s_tslab[w];
        const uint32_t n_slab = s_tslab[slab];
        for (uint32_t k = (uint32_t)lane; k < n_slab; k += 64u) {
            const uint32_t rec = s_rec[slab][k];
            const int cell = (int)(rec & 511u), i = (int)((rec >> 9) & 7u), cs = (int)(rec >> 12);
            const int cx = cell >> 6, cy = (cell >> 3) & 7, cz = cell & 7;
            uint32_t tri[3];
            const unsigned long long row = s_row[cs] >> (12 * i);  // the three edge ids of triangle i in the low nibbles
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int e = (int)((uint32_t)(row >> (4 * v)) & 15u);
                const uint32_t code = (uint32_t)(SS_MC_EDGE_CODES >> (5 * e)) & 31u;  // origin corner offsets and axis of local edge e
                const int a = (int)(code & 3u);
                const int ox = cx + (int)((code >> 4) & 1u), oy = cy + (int)((code >> 3) & 1u), oz = cz + (int)((code >> 2) & 1u);
                const int nb = ((ox >> 3) << 2) | ((oy >> 3) << 1) | (oz >> 3);
                const int p = (((ox & 7) * 8) + (oy & 7)) * 8 + (oz & 7);
                const int w = p >> 6, bit = p & 63;
                const unsigned long long bl = (bit == 0) ? 0ull : (~0ull >> (64 - bit));
                tri[v] = s_vbase[nb] + s_pref[nb][a * 8 + w] + (uint32_t)__popcll(s_mask[nb][a * 8 + w] & bl);
            }
            const size_t o = 3 * (size_t)(toff + k);
            triangles[o] = tri[0];
            triangles[o + 1] = tri[1];
            triangles[o + 2] = tri[2];
        }
    }
}

template <class R>
void ss_launch_mc_neighbours(const SSDevT<R>& P, const uint32_t* mc_xyz, uint32_t n_mc, const uint32_t* block_slot, const uint32_t* mc_slot, const uint32_t* certified, uint32_t* mc_nb, hipStream_t st) {
    if (!n_mc) return;
    hipLaunchKernelGGL(k_mc_neighbours<R>, dim3((n_mc * 8u + 255u) / 256u), dim3(256), 0, st, P, mc_xyz, n_mc, block_slot, mc_slot, certified, mc_nb);
}
template <class R>
void ss_launch_mc_count(const SSDevT<R>& P, const R* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, uint32_t n_mc,
                        unsigned long long* masks, uint32_t* vcount, uint32_t* tcount, hipStream_t st) {
    if (!n_mc) return;
    hipLaunchKernelGGL(k_mc_count<R>, dim3(n_mc), dim3(128), 0, st, P, G, mc_nb, mc_xyz, n_mc, masks, vcount, tcount);
}
template <class R>
void ss_launch_mc_emit(const SSDevT<R>& P, const R* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, const uint32_t* mc_slot,
                       uint32_t n_mc, const unsigned long long* masks, const uint32_t* vbase, const uint32_t* tbase, R* vertices,
                       unsigned long long* vkeys, uint32_t* triangles, hipStream_t st) {
    if (!n_mc) return;
    hipLaunchKernelGGL(k_mc_emit<R>, dim3(n_mc), dim3(256), 0, st, P, G, mc_nb, mc_xyz, mc_slot, n_mc, masks, vbase, tbase, vertices, vkeys,
                       triangles);
}

// =====================================================================================================
// HBM bandwidth probe (ss_measure_hbm_bandwidth): a float4 read stream or a float4 copy, grid-stride, the
// achievable rate the splat's roofline fraction is also quoted against
// =====================================================================================================
template <bool COPY>
__global__ __launch_bounds__(256) void k_stream_probe(const float4* __restrict__ in, float4* __restrict__ out, size_t n, float* __restrict__ sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        if (COPY)
            out[i] = v;
        else
            acc += v.x + v.y + v.z + v.w;
    }
    if (!COPY && acc == 123.456f) *sink = acc;  // (keeps the loads alive; never true for the zeroed buffer)
}
void ss_launch_stream_probe(bool copy, const void* in, void* out, size_t n_float4, float* sink, hipStream_t st) {
    const dim3 g(256 * 32), b(256);
    if (copy)
        hipLaunchKernelGGL(k_stream_probe<true>, g, b, 0, st, (const float4*)in, (float4*)out, n_float4, sink);
    else
        hipLaunchKernelGGL(k_stream_probe<false>, g, b, 0, st, (const float4*)in, (float4*)out, n_float4, sink);
}

// =====================================================================================================
// helpers: widen triangle indices, level-set box extraction (tests), reference decomposition statistics
// =====================================================================================================
__global__ __launch_bounds__(256) void k_widen_u32_u64(const uint32_t* __restrict__ in, size_t n, unsigned long long* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
void ss_launch_widen(const uint32_t* in, size_t n, unsigned long long* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_widen_u32_u64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
}

template <class R>
__global__ __launch_bounds__(256) void k_levelset_box(SSDevT<R> P, const R* __restrict__ G, const uint32_t* __restrict__ block_slot, int lo0,
                                                      int lo1, int lo2, int e0, int e1, int e2, R* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t tot = (size_t)e0 * e1 * e2;
    if (i >= tot) return;
    int z = (int)(i % e2), y = (int)((i / e2) % e1), x = (int)(i / ((size_t)e2 * e1));
    int gx = lo0 + x, gy = lo1 + y, gz = lo2 + z;
    R v = R(0.0);
    if (gx >= 0 && gy >= 0 && gz >= 0 && gx < P.np[0] && gy < P.np[1] && gz < P.np[2]) {
        uint32_t slot = ss_block_in_table(P, gx >> 3, gy >> 3, gz >> 3) ? block_slot[ss_block_index(P, gx >> 3, gy >> 3, gz >> 3)] : 0xFFFFFFFFu;
        if (slot != 0xFFFFFFFFu) v = G[(size_t)slot * SS_BLOCK_POINTS + (size_t)SS_BLOCK_OFFSET(gx & 7, gy & 7, gz & 7)];
    }
    out[i] = v;
}
template <class R>
void ss_launch_levelset_box(const SSDevT<R>& P, const R* G, const uint32_t* block_slot, const int lo[3], const int ext[3], R* out,
                            hipStream_t st) {
    size_t tot = (size_t)ext[0] * ext[1] * ext[2];
    if (!tot) return;
    hipLaunchKernelGGL(k_levelset_box<R>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, P, G, block_slot, lo[0], lo[1], lo[2], ext[0],
                       ext[1], ext[2], out);
}

// =====================================================================================================
// Fused scans of the host flow (ss_prims.h: one dispatch each; the functors below are the kernels that used to run before and
// after a library scan in rounds 1-3).  `state`: zeroed scan state (ss_scan_state_words); `mail`: where the total is posted for the host.
// =====================================================================================================
// sorted order -> payload, and the run starts of the cell table in the same pass: first[c] = ~(position of the first entry of cell c),
// first[ncells] = ~n, 0 = no entry (the table is preset to 0); ss_launch_cell_table_scan turns it into cell_start
template <class R, bool OWNED>
__global__ __launch_bounds__(256) void k_sorted_gather_runs(SSDevT<R> P, uint32_t n, const R* __restrict__ xyz, const uint32_t* __restrict__ perm, ss_pos<R>* __restrict__ pos_sorted,
                                                            const uint32_t* __restrict__ sorted_keys, uint32_t ncells, uint32_t* __restrict__ first,
                                                            const uint32_t* __restrict__ occ_sub, uint8_t* __restrict__ owned) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n) return;
    if (p == n) {
        first[ncells] = ~n;
        return;
    }
    const size_t i = perm[p];
    const R x3[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    pos_sorted[p] = ss_make_pos<R>(x3[0], x3[1], x3[2]);
    const uint32_t k = sorted_keys[p];
    if (p == 0 || sorted_keys[p - 1] != k) first[k] = ~p;
    if constexpr (OWNED) {
        // subdomain copies: is this the copy whose density its subdomain computes, i.e. does it lie inside the subdomain's half-open AABB
        // (is_inside, dense_subdomains.rs:567-576, aabb.rs:220-222)?  The others are ghosts, their density comes from another subdomain.
        const uint32_t ctot = (uint32_t)(P.sc[0] * P.sc[1] * P.sc[2]);
        const uint32_t flat = occ_sub[k / ctot];
        const int s3[3] = {(int)(flat / ((uint32_t)P.ns[2] * (uint32_t)P.ns[1])), (int)((flat / (uint32_t)P.ns[2]) % (uint32_t)P.ns[1]), (int)(flat % (uint32_t)P.ns[2])};
        uint8_t f = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const R lo = P.gmin[d] + (R)s3[d] * P.sub_size;
            const R hi = P.gmin[d] + (R)(s3[d] + 1) * P.sub_size;
            if (!(x3[d] >= lo && x3[d] < hi)) f = 0;
        }
        owned[p] = f;
    }
}
template <class R>
void ss_launch_sorted_gather_runs(const SSDevT<R>& P, uint32_t n, const R* xyz, const uint32_t* perm, ss_pos<R>* pos_sorted, const uint32_t* sorted_keys, uint32_t ncells, uint32_t* first,
                                  const uint32_t* occ_sub, uint8_t* owned, hipStream_t st) {
    if (owned)
        hipLaunchKernelGGL((k_sorted_gather_runs<R, true>), dim3((n + 1u + 255u) / 256u), dim3(256), 0, st, P, n, xyz, perm, pos_sorted, sorted_keys, ncells, first, occ_sub, owned);
    else
        hipLaunchKernelGGL((k_sorted_gather_runs<R, false>), dim3((n + 1u + 255u) / 256u), dim3(256), 0, st, P, n, xyz, perm, pos_sorted, sorted_keys, ncells, first, occ_sub, owned);
}

// cell_start[c] = first entry of cell c or of the next non-empty cell (n behind the last): a running maximum of the complemented
// run starts from the END of the table (an empty cell is 0, the identity)
struct SSCellTableIn {
    const uint32_t* first;
    uint32_t ncells;
    __device__ uint32_t operator()(uint32_t j) const { return first[ncells - j]; }
};
struct SSCellTableOut {
    uint32_t* cell_start;
    uint32_t ncells;
    __device__ void operator()(uint32_t j, uint32_t x, uint32_t excl) const { cell_start[ncells - j] = ~(x > excl ? x : excl); }
};
void ss_launch_cell_table_scan(const uint32_t* first, uint32_t ncells, uint32_t* cell_start, uint32_t* state, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpMax>(SSCellTableIn{first, ncells}, SSCellTableOut{cell_start, ncells}, ncells + 1u, state, (uint32_t*)nullptr, SSMailSlot{}, st);
}

// membership count per particle as the scan's input: copy_offset[i] = copies of the particles before i; the flags of the
// subdomains with particles are set on the way
template <class R>
struct SSClassifyIn {
    SSDevT<R> P;
    const R* xyz;
    uint32_t* sub_flag;
    __device__ uint32_t operator()(uint32_t i) const {
        const R p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
        uint32_t m = 0;
        const SSDevT<R>& Q = P;
        uint32_t* flags = sub_flag;
        ss_for_each_member_subdomain(Q, p, [&](int sx, int sy, int sz) {
            ++m;
            uint32_t* f = flags + ((size_t)sx * Q.ns[1] + sy) * Q.ns[2] + sz;
            if (!*f) *f = 1u;
        });
        return m;
    }
};
struct SSStoreExcl {
    uint32_t* out;
    __device__ void operator()(uint32_t i, uint32_t, uint32_t excl) const { out[i] = excl; }
};
template <class R>
void ss_launch_classify_scan(const SSDevT<R>& P, const R* xyz, uint32_t* copy_offset, uint32_t* sub_flag, uint32_t* state, SSMailSlot mail, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpPlus>(SSClassifyIn<R>{P, xyz, sub_flag}, SSStoreExcl{copy_offset}, P.n, state, (uint32_t*)nullptr, mail, st);
}

// flags -> ranks and the compacted list of the set entries (occupied subdomains; blocks to complete)
struct SSFlagIn {
    const uint32_t* flag;
    __device__ uint32_t operator()(uint32_t i) const { return flag[i] ? 1u : 0u; }
};
struct SSRankListOut {
    uint32_t* rank;  // may be null
    uint32_t* list;
    __device__ void operator()(uint32_t i, uint32_t f, uint32_t excl) const {
        if (rank) rank[i] = excl;
        if (f) list[excl] = i;
    }
};
void ss_launch_flag_scan(const uint32_t* flag, uint32_t n, uint32_t* rank, uint32_t* list, uint32_t* total_dev, uint32_t* state, SSMailSlot mail, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpPlus>(SSFlagIn{flag}, SSRankListOut{rank, list}, n, state, total_dev, mail, st);
}

// the copies whose density their subdomain computes (flags from k_sorted_gather_runs), compacted in cell order; the count stays on the device
struct SSByteFlagIn {
    const uint8_t* flag;
    __device__ uint32_t operator()(uint32_t i) const { return (uint32_t)flag[i]; }
};
void ss_launch_owned_scan(uint32_t n_copies, const uint8_t* owned, uint32_t* own_list, uint32_t* n_owned_dev, uint32_t* state, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpPlus>(SSByteFlagIn{owned}, SSRankListOut{nullptr, own_list}, n_copies, state, n_owned_dev, SSMailSlot{}, st);
}

// active level-set blocks: flags -> list, slot table and block coordinates in one pass; entries beyond
// `cap` are not written (the host re-runs with larger buffers when the total exceeds it)
template <class R>
struct SSBlockListOut {
    SSDevT<R> P;
    uint32_t cap;
    uint32_t* list;
    uint32_t* slot;
    uint32_t* xyz;
    __device__ void operator()(uint32_t b, uint32_t f, uint32_t excl) const {
        if (!f) {
            slot[b] = 0xFFFFFFFFu;
            return;
        }
        slot[b] = excl;
        if (excl < cap) {
            if (list) list[excl] = b;
            int bx, by, bz;
            ss_block_of_index(P, b, &bx, &by, &bz);
            xyz[3 * (size_t)excl + 0] = (uint32_t)bx;
            xyz[3 * (size_t)excl + 1] = (uint32_t)by;
            xyz[3 * (size_t)excl + 2] = (uint32_t)bz;
        }
    }
};
template <class R>
void ss_launch_active_blocks_scan(const SSDevT<R>& P, const uint32_t* block_flag, uint32_t nblocks, uint32_t cap, uint32_t* list, uint32_t* slot, uint32_t* xyz, uint32_t* state,
                                  SSMailSlot mail, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpPlus>(SSFlagIn{block_flag}, SSBlockListOut<R>{P, cap, list, slot, xyz}, nblocks, state, (uint32_t*)nullptr, mail, st);
}

// marching-cubes blocks: the flag (do the eight level-set blocks b + {0,1}^3 hold values on both sides of the threshold?) computed as the scan's input, list / slot table / coordinates as its output
template <class R>
struct SSMcFlagIn {
    SSDevT<R> P;
    const uint32_t* block_slot;
    const ss_real2<R>* blk_minmax;
    __device__ uint32_t operator()(uint32_t b) const {
        int bx, by, bz;
        ss_block_of_index(P, b, &bx, &by, &bz);
        if (bx < P.blk_lo[0] || by < P.blk_lo[1] || bz < P.blk_lo[2] || bx > P.blk_hi[0] || by > P.blk_hi[1] || bz > P.blk_hi[2]) return 0u;
        bool any_in = false, any_out = false;
        for (int dx = 0; dx <= 1; ++dx)
            for (int dy = 0; dy <= 1; ++dy)
                for (int dz = 0; dz <= 1; ++dz) {
                    const int x = bx + dx, y = by + dy, z = bz + dz;
                    R mn = R(0.0), mx = R(0.0);
                    if (ss_block_in_table(P, x, y, z)) {
                        const uint32_t slot = block_slot[ss_block_index(P, x, y, z)];
                        if (slot != 0xFFFFFFFFu) {
                            const ss_real2<R> mm = blk_minmax[slot];
                            mn = mm.x;
                            mx = mm.y;
                        }
                    }
                    any_in = any_in || (mx > P.threshold);
                    any_out = any_out || !(mn > P.threshold);
                }
        return (any_in && any_out) ? 1u : 0u;
    }
};
template <class R>
void ss_launch_mc_blocks_scan(const SSDevT<R>& P, const uint32_t* block_slot, const ss_real2<R>* blk_minmax, uint32_t nblocks, uint32_t cap, uint32_t* mc_list, uint32_t* mc_slot,
                              uint32_t* mc_xyz, uint32_t* state, SSMailSlot mail, hipStream_t st) {
    ss_chained_scan<uint32_t, SSOpPlus>(SSMcFlagIn<R>{P, block_slot, blk_minmax}, SSBlockListOut<R>{P, cap, mc_list, mc_slot, mc_xyz}, nblocks, state, (uint32_t*)nullptr, mail, st);
}

// vertex and triangle offsets of the marching-cubes blocks in ONE scan: the two counts packed into one 64-bit value, 31 bits each -- the scan's
// status words keep bits 63:62 for their flags (ss_prims.h), so the running (vertex | triangle << 31) totals have to stay below 2^62.  The host
// guarantees that from the number of blocks (SS_MC_MAX_TRI_PER_BLOCK); larger jobs take two 64-bit scans (`split`).
struct SSMcCountsIn {
    const uint32_t* vcount;
    const uint32_t* tcount;
    __device__ unsigned long long operator()(uint32_t i) const { return (unsigned long long)vcount[i] | ((unsigned long long)tcount[i] << 31); }
};
struct SSMcOffsetsOut {
    uint32_t* vbase;
    uint32_t* tbase;
    uint32_t n;
    __device__ void operator()(uint32_t i, unsigned long long x, unsigned long long excl) const {
        vbase[i] = (uint32_t)(excl & 0x7FFFFFFFull);
        tbase[i] = (uint32_t)(excl >> 31);
        if (i + 1u == n) {  // entry n: the totals (the emit kernel reads base[m + 1] of the last block)
            const unsigned long long t = excl + x;
            vbase[n] = (uint32_t)(t & 0x7FFFFFFFull);
            tbase[n] = (uint32_t)(t >> 31);
        }
    }
};
struct SSStoreExcl32 {  // (totals of the split form stay below 2^32: the host checks the posted 64-bit totals before anybody reads the bases)
    uint32_t* out;
    uint32_t n;
    __device__ void operator()(uint32_t i, unsigned long long x, unsigned long long excl) const {
        out[i] = (uint32_t)excl;
        if (i + 1u == n) out[n] = (uint32_t)(excl + x);
    }
};
struct SSWidenIn;
void ss_launch_mc_offsets_scan(const uint32_t* vcount, const uint32_t* tcount, uint32_t n_mc, uint32_t* vbase, uint32_t* tbase, uint32_t* state, uint32_t* state2, SSMailSlot mail,
                               SSMailSlot mail2, hipStream_t st) {
    if (!state2) {  // packed: mail = vertices | triangles << 31
        ss_chained_scan<unsigned long long, SSOpPlus>(SSMcCountsIn{vcount, tcount}, SSMcOffsetsOut{vbase, tbase, n_mc}, n_mc, state, (unsigned long long*)nullptr, mail, st);
        return;
    }
    struct Widen {
        const uint32_t* v;
        __device__ unsigned long long operator()(uint32_t i) const { return (unsigned long long)v[i]; }
    };
    ss_chained_scan<unsigned long long, SSOpPlus>(Widen{vcount}, SSStoreExcl32{vbase, n_mc}, n_mc, state, (unsigned long long*)nullptr, mail, st);    // mail = vertices
    ss_chained_scan<unsigned long long, SSOpPlus>(Widen{tcount}, SSStoreExcl32{tbase, n_mc}, n_mc, state2, (unsigned long long*)nullptr, mail2, st);  // mail2 = triangles
}

// 64-bit offsets of the tiles in the arena from the 32-bit bounds
struct SSWidenIn {
    const uint32_t* v;
    __device__ unsigned long long operator()(uint32_t i) const { return (unsigned long long)v[i]; }
};
struct SSStoreExcl64 {
    unsigned long long* out;
    uint32_t n;
    __device__ void operator()(uint32_t i, unsigned long long x, unsigned long long excl) const {
        out[i] = excl;
        if (i + 1u == n) out[n] = excl + x;
    }
};
void ss_launch_tile_offsets_scan(const uint32_t* bound, uint32_t n, unsigned long long* off, uint32_t* state, SSMailSlot mail, hipStream_t st) {
    ss_chained_scan<unsigned long long, SSOpPlus>(SSWidenIn{bound}, SSStoreExcl64{off, n}, n, state, (unsigned long long*)nullptr, mail, st);
}

// a count that lives on the device (the length of a list built with atomics), posted for the host

template void ss_launch_sorted_gather_runs<float>(const SSDevT<float>&, uint32_t, const float*, const uint32_t*, ss_pos<float>*, const uint32_t*, uint32_t, uint32_t*, const uint32_t*, uint8_t*, hipStream_t);
template void ss_launch_sorted_gather_runs<double>(const SSDevT<double>&, uint32_t, const double*, const uint32_t*, ss_pos<double>*, const uint32_t*, uint32_t, uint32_t*, const uint32_t*, uint8_t*, hipStream_t);
template void ss_launch_classify_scan<float>(const SSDevT<float>&, const float*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);
template void ss_launch_classify_scan<double>(const SSDevT<double>&, const double*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);
template void ss_launch_active_blocks_scan<float>(const SSDevT<float>&, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);
template void ss_launch_active_blocks_scan<double>(const SSDevT<double>&, const uint32_t*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);
template void ss_launch_mc_blocks_scan<float>(const SSDevT<float>&, const uint32_t*, const ss_real2<float>*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);
template void ss_launch_mc_blocks_scan<double>(const SSDevT<double>&, const uint32_t*, const ss_real2<double>*, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, SSMailSlot, hipStream_t);

// ---- explicit instantiations of the launch wrappers (f32: reconstruct_surface::<i64,f32>, f64: ::<i64,f64>) ----
template void ss_launch_aabb<float>(const float* d_xyz, uint32_t n, float* d_partial, float* d_out6, SSMailSlot mail, hipStream_t st);
template void ss_launch_aabb<double>(const double* d_xyz, uint32_t n, double* d_partial, double* d_out6, SSMailSlot mail, hipStream_t st);
template void ss_launch_inside_flags<float>(const float* d_xyz, uint32_t n, const float amin[3], const float amax[3], uint8_t* f8, uint32_t* f32, hipStream_t st);
template void ss_launch_inside_flags<double>(const double* d_xyz, uint32_t n, const double amin[3], const double amax[3], uint8_t* f8, uint32_t* f32, hipStream_t st);
template void ss_launch_compact_xyz<float>(const float* d_xyz, uint32_t n, const uint32_t* f32, const uint32_t* offs, float* out, hipStream_t st);
template void ss_launch_compact_xyz<double>(const double* d_xyz, uint32_t n, const uint32_t* f32, const uint32_t* offs, double* out, hipStream_t st);
template void ss_launch_cell_keys<float>(const SSDevT<float>& P, const float* d_xyz, uint32_t* keys, uint32_t* vals, hipStream_t st);
template void ss_launch_cell_keys<double>(const SSDevT<double>& P, const double* d_xyz, uint32_t* keys, uint32_t* vals, hipStream_t st);
template void ss_launch_emit_copies<float>(const SSDevT<float>& P, const float* xyz, const uint32_t* copy_offset, const uint32_t* occ_rank, uint32_t* keys, uint32_t* vals, hipStream_t st);
template void ss_launch_emit_copies<double>(const SSDevT<double>& P, const double* xyz, const uint32_t* copy_offset, const uint32_t* occ_rank, uint32_t* keys, uint32_t* vals, hipStream_t st);
template void ss_launch_density_sub<float>(const SSDevT<float>& P, uint32_t n_copies, const ss_pos<float>* cpos, const uint32_t* cidx, const uint32_t* ckey, const uint32_t* cell_start, const uint32_t* occ_sub, float* rho, int mode, uint32_t* nb_count, const unsigned long long* nb_ptr, uint32_t* nb_idx, bool fast_div, const uint32_t* owned_list, const uint32_t* n_owned_dev, uint32_t n_owned_bound, hipStream_t st);
template void ss_launch_density_sub<double>(const SSDevT<double>& P, uint32_t n_copies, const ss_pos<double>* cpos, const uint32_t* cidx, const uint32_t* ckey, const uint32_t* cell_start, const uint32_t* occ_sub, double* rho, int mode, uint32_t* nb_count, const unsigned long long* nb_ptr, uint32_t* nb_idx, bool fast_div, const uint32_t* owned_list, const uint32_t* n_owned_dev, uint32_t n_owned_bound, hipStream_t st);
template void ss_launch_make_posvol<float>(const SSDevT<float>& P, const ss_pos<float>* pos_sorted, const uint32_t* perm, const float* rho, ss_real4<float>* posvol, ss_real4<float>* posvol_by_index, hipStream_t st);
template void ss_launch_make_posvol<double>(const SSDevT<double>& P, const ss_pos<double>* pos_sorted, const uint32_t* perm, const double* rho, ss_real4<double>* posvol, ss_real4<double>* posvol_by_index, hipStream_t st);
template void ss_launch_mark_blocks<float>(const SSDevT<float>& P, const uint32_t* cell_start, uint32_t ncells, uint32_t* block_flag, hipStream_t st);
template void ss_launch_mark_blocks<double>(const SSDevT<double>& P, const uint32_t* cell_start, uint32_t ncells, uint32_t* block_flag, hipStream_t st);
template void ss_launch_splat_bounds<float>(const SSDevT<float>& P, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* counts, uint32_t* bound, hipStream_t st);
template void ss_launch_splat_gather<float>(const SSDevT<float>& P, const ss_real4<float>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const unsigned long long* tile_off, ss_real4<float>* arena, uint32_t* arena_idx, uint32_t* counts, uint32_t* large_flag, hipStream_t st);
template void ss_launch_splat_gather_large<float>(const SSDevT<float>& P, const ss_real4<float>* posvol, const ss_real4<float>* posvol_by_index, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, const uint32_t* large_list, const uint32_t* n_large_dev, const uint32_t* counts, const unsigned long long* tile_off, ss_real4<float>* arena, uint32_t* arena_idx, hipStream_t st);
template void ss_launch_splat_bounds<double>(const SSDevT<double>& P, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* counts, uint32_t* bound, hipStream_t st);
template void ss_launch_splat_gather<double>(const SSDevT<double>& P, const ss_real4<double>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, uint32_t n_active, const unsigned long long* tile_off, ss_real4<double>* arena, uint32_t* arena_idx, uint32_t* counts, uint32_t* large_flag, hipStream_t st);
template void ss_launch_splat_gather_large<double>(const SSDevT<double>& P, const ss_real4<double>* posvol, const ss_real4<double>* posvol_by_index, const uint32_t* perm, const uint32_t* cell_start, const uint32_t* active_xyz, const uint32_t* large_list, const uint32_t* n_large_dev, const uint32_t* counts, const unsigned long long* tile_off, ss_real4<double>* arena, uint32_t* arena_idx, hipStream_t st);
template void ss_launch_splat_fused<float>(const SSDevT<float>& P, const ss_real4<float>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint2* row_tab, const uint32_t* active_xyz, uint32_t n_active, float* G, ss_real2<float>* blk_minmax, uint32_t* trunc, bool full_levelset, const uint32_t* list, const uint32_t* n_list_dev, const uint32_t* redo_mask, unsigned long long* facebits, uint32_t* counts, uint32_t* big, hipStream_t st);
template void ss_launch_splat_accumulate_big<float>(const SSDevT<float>& P, const ss_real4<float>* arena, const uint32_t* arena_idx, const unsigned long long* tile_off, const uint32_t* counts, const uint32_t* active_xyz, float* G, ss_real2<float>* blk_minmax, uint32_t* trunc, bool full_levelset, bool second_pass, bool exact_first, const uint32_t* redo_mask, unsigned long long* facebits, const uint32_t* big, uint32_t* err, hipStream_t st);
template void ss_launch_select_redo<float>(const SSDevT<float>& P, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* block_slot, const uint32_t* trunc, const unsigned long long* facebits, uint32_t* redo_mask, const uint32_t* counts, unsigned long long* stats, uint32_t* big, hipStream_t st);
template void ss_launch_splat_fused<double>(const SSDevT<double>& P, const ss_real4<double>* posvol, const uint32_t* perm, const uint32_t* cell_start, const uint2* row_tab, const uint32_t* active_xyz, uint32_t n_active, double* G, ss_real2<double>* blk_minmax, uint32_t* trunc, bool full_levelset, const uint32_t* list, const uint32_t* n_list_dev, const uint32_t* redo_mask, unsigned long long* facebits, uint32_t* counts, uint32_t* big, hipStream_t st);
template void ss_launch_splat_accumulate_big<double>(const SSDevT<double>& P, const ss_real4<double>* arena, const uint32_t* arena_idx, const unsigned long long* tile_off, const uint32_t* counts, const uint32_t* active_xyz, double* G, ss_real2<double>* blk_minmax, uint32_t* trunc, bool full_levelset, bool second_pass, bool exact_first, const uint32_t* redo_mask, unsigned long long* facebits, const uint32_t* big, uint32_t* err, hipStream_t st);
template void ss_launch_select_redo<double>(const SSDevT<double>& P, const uint32_t* active_xyz, uint32_t n_active, const uint32_t* block_slot, const uint32_t* trunc, const unsigned long long* facebits, uint32_t* redo_mask, const uint32_t* counts, unsigned long long* stats, uint32_t* big, hipStream_t st);
template void ss_launch_mc_neighbours<float>(const SSDevT<float>& P, const uint32_t* mc_xyz, uint32_t n_mc, const uint32_t* block_slot, const uint32_t* mc_slot, const uint32_t* certified, uint32_t* mc_nb, hipStream_t st);
template void ss_launch_mc_count<float>(const SSDevT<float>& P, const float* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, uint32_t n_mc, unsigned long long* masks, uint32_t* vcount, uint32_t* tcount, hipStream_t st);
template void ss_launch_mc_neighbours<double>(const SSDevT<double>& P, const uint32_t* mc_xyz, uint32_t n_mc, const uint32_t* block_slot, const uint32_t* mc_slot, const uint32_t* certified, uint32_t* mc_nb, hipStream_t st);
template void ss_launch_mc_count<double>(const SSDevT<double>& P, const double* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, uint32_t n_mc, unsigned long long* masks, uint32_t* vcount, uint32_t* tcount, hipStream_t st);
template void ss_launch_mc_emit<float>(const SSDevT<float>& P, const float* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, const uint32_t* mc_slot, uint32_t n_mc, const unsigned long long* masks, const uint32_t* vbase, const uint32_t* tbase, float* vertices, unsigned long long* vkeys, uint32_t* triangles, hipStream_t st);
template void ss_launch_mc_emit<double>(const SSDevT<double>& P, const double* G, const uint32_t* mc_nb, const uint32_t* mc_xyz, const uint32_t* mc_slot, uint32_t n_mc, const unsigned long long* masks, const uint32_t* vbase, const uint32_t* tbase, double* vertices, unsigned long long* vkeys, uint32_t* triangles, hipStream_t st);
template void ss_launch_levelset_box<float>(const SSDevT<float>& P, const float* G, const uint32_t* block_slot, const int lo[3], const int ext[3], float* out, hipStream_t st);
template void ss_launch_levelset_box<double>(const SSDevT<double>& P, const double* G, const uint32_t* block_slot, const int lo[3], const int ext[3], double* out, hipStream_t st);
