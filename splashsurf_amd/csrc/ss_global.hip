// ss_global.hip -- gfx950 kernels of the reference's GLOBAL (non-decomposed) strategy, SURVEY rows A14/A15.
// Citations are relative to /root/reference/splashsurf_lib/src/.
//
//   neighborhood_search.rs:148-230 (sequential spatial hashing) + density_map.rs:113-186  -> k_g_cell_keys, ss_radix_sort_pairs (ss_prims.hip), k_g_density
//   density_map.rs:364-412, 582-737 (SparseDensityMapGenerator, sequential)               -> k_g_chunk_boxes, k_g_levelset
//   marching_cubes/narrow_band_extraction.rs:8-219 + triangulation.rs:23-95                -> k_g_edge_masks, k_g_cell_count,
//                                                                                            k_g_emit_vertices, k_g_emit_triangles
//
// Every grid point is owned by one thread that gathers its contributions in ascending particle index, the
// order of the reference's sequential loop (density_map.rs:389-396); the per-particle running offsets
// dx += cell_size of particle_support_loop (density_map.rs:693-735) are reproduced addition by addition.
// This strategy serves small domains (it is what `auto_disable` selects for <= 1.2 n cells per dimension), so
// the kernels favour exactness and simplicity over throughput; large inputs belong to the subdomain path.
#include <climits>

#include "ss_global.h"

__constant__ int8_t g_mc_table[256][16] = {
#include "mc_table.inc"
};
// uniform_grid.rs:825-834
__constant__ int8_t g_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// uniform_grid.rs:856-869: local edge -> (origin corner, axis)
__constant__ int8_t g_edge[12][2] = {{0, 0}, {1, 1}, {3, 0}, {0, 1}, {4, 0}, {5, 1}, {7, 0}, {4, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}};

// enclosing_cell (uniform_grid.rs:444-451) of the search grid
template <class R>
__device__ inline void ssg_search_cell(const SSGlobT<R>& P, const R* p, int c[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = (int)ss_floor((p[d] - P.smin[d]) / P.h);
}

// =====================================================================================================
// cell -> particles map (neighborhood_search.rs:655-676): keys for a stable sort, which yields the ascending
// particle index per cell of the reference's sequential push
// =====================================================================================================
template <class R>
__global__ __launch_bounds__(256) void k_g_cell_keys(SSGlobT<R> P, const R* __restrict__ xyz, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                     uint32_t* __restrict__ cell_count, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const R p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    int c[3];
    ssg_search_cell(P, p, c);
    uint32_t key = 0;
    if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= P.snc[0] || c[1] >= P.snc[1] || c[2] >= P.snc[2]) {
        atomicOr(err, 1u);  // grid.get_cell(..).unwrap() panics in the reference (:664)
    } else {
        key = (uint32_t)((c[0] * P.snc[1] + c[1]) * P.snc[2] + c[2]);
        atomicAdd(&cell_count[key], 1u);
    }
    keys[i] = key;
    vals[i] = i;
}
template <class R>
void ssg_launch_cell_keys(const SSGlobT<R>& P, const R* xyz, uint32_t* keys, uint32_t* vals, uint32_t* cell_count, uint32_t* err, hipStream_t st) {
    if (!P.n) return;
    hipLaunchKernelGGL(k_g_cell_keys<R>, dim3((P.n + 255) / 256), dim3(256), 0, st, P, xyz, keys, vals, cell_count, err);
}

// =====================================================================================================
// neighbour lists + densities, one thread per particle (neighborhood_search.rs:186-227, density_map.rs:165-184)
// =====================================================================================================
template <class R, int MODE>
__global__ __launch_bounds__(256) void k_g_density(SSGlobT<R> P, const R* __restrict__ xyz, const uint32_t* __restrict__ cell_start,
                                                   const uint32_t* __restrict__ items, R* __restrict__ rho, uint32_t* __restrict__ nb_count,
                                                   const unsigned long long* __restrict__ nb_ptr, uint32_t* __restrict__ nb_idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const R pi[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    int ci[3];
    ssg_search_cell(P, pi, ci);
    R acc = P.w0;  // density_map.rs:173
    uint32_t cnt = 0;
    unsigned long long wr = (MODE == 2) ? nb_ptr[i] : 0ull;
    // cells_adjacent_to_cell (26 cells, steps in (x,y,z) lexicographic order, uniform_grid.rs:614-643) chained with the own cell
    for (int ph = 0; ph < 2; ++ph)
        for (int sx = -1; sx <= 1; ++sx)
            for (int sy = -1; sy <= 1; ++sy)
                for (int sz = -1; sz <= 1; ++sz) {
                    const bool center = (sx == 0 && sy == 0 && sz == 0);
                    if ((ph == 0) == center) continue;
                    const int cx = ci[0] + sx, cy = ci[1] + sy, cz = ci[2] + sz;
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= P.snc[0] || cy >= P.snc[1] || cz >= P.snc[2]) continue;
                    const uint32_t f = (uint32_t)((cx * P.snc[1] + cy) * P.snc[2] + cz);
                    const uint32_t b = cell_start[f], e = cell_start[f + 1];
                    for (uint32_t q = b; q < e; ++q) {
                        const uint32_t j = items[q];
                        if (j == i) continue;  // :216-218
                        const R dx = xyz[3 * (size_t)j] - pi[0], dy = xyz[3 * (size_t)j + 1] - pi[1], dz = xyz[3 * (size_t)j + 2] - pi[2];
                        const R d2 = dx * dx + dy * dy + dz * dz;  // nalgebra norm_squared
                        if (d2 < P.h2) {                            // :221
                            if (MODE == 0) acc += ss_kernel_evaluate<R>(ss_sqrt(d2), P.h, P.sigma);  // density_map.rs:176-180
                            if (MODE == 2) nb_idx[wr++] = j;
                            ++cnt;
                        }
                    }
                }
    if (MODE == 0) {
        rho[i] = acc * P.mass;  // density_map.rs:182
        nb_count[i] = cnt;
    }
}
template <class R>
void ssg_launch_density(const SSGlobT<R>& P, const R* xyz, const uint32_t* cell_start, const uint32_t* cell_items, R* rho, int mode, uint32_t* nb_count,
                        const unsigned long long* nb_ptr, uint32_t* nb_idx, hipStream_t st) {
    if (!P.n) return;
    const dim3 g((P.n + 255) / 256), b(256);
    if (mode == 0)
        hipLaunchKernelGGL((k_g_density<R, 0>), g, b, 0, st, P, xyz, cell_start, cell_items, rho, nb_count, nb_ptr, nb_idx);
    else
        hipLaunchKernelGGL((k_g_density<R, 2>), g, b, 0, st, P, xyz, cell_start, cell_items, rho, nb_count, nb_ptr, nb_idx);
}

// =====================================================================================================
// level set (SparseDensityMapGenerator::compute_particle_density_contribution, density_map.rs:642-735)
// =====================================================================================================
// stencil box of a particle: [lo, lo + supported) per axis; false if the particle is skipped (:648-651)
template <class R>
__device__ inline bool ssg_stencil(const SSGlobT<R>& P, const R p[3], int lo[3]) {
    if (!(p[0] >= P.amin[0] && p[1] >= P.amin[1] && p[2] >= P.amin[2] && p[0] < P.amax[0] && p[1] < P.amax[1] && p[2] < P.amax[2])) return false;
#pragma unroll
    for (int d = 0; d < 3; ++d) lo[d] = (int)ss_floor((p[d] - P.gmin[d]) / P.cs) - P.half_cells;  // enclosing_cell - half_supported_cells
    return true;
}

template <class R>
__global__ __launch_bounds__(SS_GCHUNK) void k_g_chunk_boxes(SSGlobT<R> P, const R* __restrict__ xyz, int* __restrict__ boxes) {
    __shared__ int s_lo[3][SS_GCHUNK];
    __shared__ int s_hi[3][SS_GCHUNK];
    const uint32_t i = blockIdx.x * SS_GCHUNK + threadIdx.x;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (i < P.n) {
        const R p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
        int l[3];
        if (ssg_stencil(P, p, l))
            for (int d = 0; d < 3; ++d) {
                lo[d] = l[d];
                hi[d] = l[d] + P.supported;
            }
    }
    for (int d = 0; d < 3; ++d) {
        s_lo[d][threadIdx.x] = lo[d];
        s_hi[d][threadIdx.x] = hi[d];
    }
    __syncthreads();
    for (int s = SS_GCHUNK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int d = 0; d < 3; ++d) {
                s_lo[d][threadIdx.x] = min(s_lo[d][threadIdx.x], s_lo[d][threadIdx.x + s]);
                s_hi[d][threadIdx.x] = max(s_hi[d][threadIdx.x], s_hi[d][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int d = 0; d < 3; ++d) {
            boxes[6 * (size_t)blockIdx.x + d] = s_lo[d][0];
            boxes[6 * (size_t)blockIdx.x + 3 + d] = s_hi[d][0];
        }
}
template <class R>
void ssg_launch_chunk_boxes(const SSGlobT<R>& P, const R* xyz, int* boxes, hipStream_t st) {
    if (!P.n) return;
    hipLaunchKernelGGL(k_g_chunk_boxes<R>, dim3((P.n + SS_GCHUNK - 1) / SS_GCHUNK), dim3(SS_GCHUNK), 0, st, P, xyz, boxes);
}

// One 512-thread workgroup per 8^3 tile of grid points; thread (u,v,w) owns one point.  Particles are
// examined SS_GCHUNK at a time in ascending index; those whose stencil box meets the tile are compacted
// (order preserving) into LDS together with their squared running offsets for the tile's 8 coordinates per
// axis -- dx after k additions of cell_size, exactly the sequence of density_map.rs:694-716 -- then every
// thread adds the chunk's contributions in order.
template <class R>
__global__ __launch_bounds__(512) void k_g_levelset(SSGlobT<R> P, const R* __restrict__ xyz, const R* __restrict__ rho, const int* __restrict__ boxes,
                                                    R* __restrict__ G) {
    __shared__ R s_sq[SS_GCHUNK][3][SS_GTILE];  // squared offsets, +inf where the point is outside the particle's stencil
    __shared__ R s_vol[SS_GCHUNK];
    __shared__ int s_wcount[SS_GCHUNK / 64];
    const int tid = (int)threadIdx.x;
    const int tiles_z = (P.np[2] + SS_GTILE - 1) / SS_GTILE, tiles_y = (P.np[1] + SS_GTILE - 1) / SS_GTILE;
    const int tz = (int)(blockIdx.x % (unsigned)tiles_z), ty = (int)((blockIdx.x / (unsigned)tiles_z) % (unsigned)tiles_y),
              tx = (int)(blockIdx.x / ((unsigned)tiles_z * (unsigned)tiles_y));
    const int t0[3] = {tx * SS_GTILE, ty * SS_GTILE, tz * SS_GTILE};
    const int u = tid >> 6, v = (tid >> 3) & 7, w = tid & 7;
    const int gi = t0[0] + u, gj = t0[1] + v, gk = t0[2] + w;
    const bool live = gi < P.np[0] && gj < P.np[1] && gk < P.np[2];
    const R inf = std::numeric_limits<R>::infinity();
    R acc = R(0.0);  // *entry(..).or_insert(0) += contribution (:722-726)
    const uint32_t n_chunks = (P.n + SS_GCHUNK - 1) / SS_GCHUNK;
    for (uint32_t ch = 0; ch < n_chunks; ++ch) {
        // chunk-level rejection (uniform across the workgroup)
        {
            const int* bx = boxes + 6 * (size_t)ch;
            bool hit = true;
#pragma unroll
            for (int d = 0; d < 3; ++d) hit = hit && bx[d] < t0[d] + SS_GTILE && bx[3 + d] > t0[d];
            if (!hit) continue;
        }
        // ---- candidate test + order-preserving compaction (threads 0..SS_GCHUNK-1 examine one particle each) ----
        bool cand = false;
        R p[3] = {R(0), R(0), R(0)};
        int lo[3] = {0, 0, 0};
        const uint32_t a = ch * SS_GCHUNK + (uint32_t)tid;
        if (tid < SS_GCHUNK && a < P.n) {
            p[0] = xyz[3 * (size_t)a];
            p[1] = xyz[3 * (size_t)a + 1];
            p[2] = xyz[3 * (size_t)a + 2];
            if (ssg_stencil(P, p, lo)) {
                cand = true;
#pragma unroll
                for (int d = 0; d < 3; ++d) cand = cand && lo[d] < t0[d] + SS_GTILE && lo[d] + P.supported > t0[d];
            }
        }
        const unsigned long long bal = __ballot(cand);
        const int lane = tid & 63, wv = tid >> 6;
        if (tid < SS_GCHUNK && lane == 0) s_wcount[wv] = __popcll(bal);
        __syncthreads();
        int n_cand = 0, base = 0;
#pragma unroll
        for (int q = 0; q < SS_GCHUNK / 64; ++q) {
            if (q < wv) base += s_wcount[q];
            n_cand += s_wcount[q];
        }
        if (cand) {
            const int slot = base + __popcll(bal & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            s_vol[slot] = P.mass / rho[a];  // particle_volume (:688)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#pragma unroll
                for (int q = 0; q < SS_GTILE; ++q) s_sq[slot][d][q] = inf;
                // min_supported_point[d] - particle[d] - cell_size, then += cell_size per step (:694-716)
                const R mp = P.gmin[d] + (R)lo[d] * P.cs;  // point_coordinates_array (uniform_grid.rs:418-431)
                R run = mp - p[d] - P.cs;
                for (int i = lo[d]; i < lo[d] + P.supported; ++i) {
                    run += P.cs;
                    const int q = i - t0[d];
                    if (q >= 0 && q < SS_GTILE) s_sq[slot][d][q] = run * run;
                }
            }
        }
        __syncthreads();
        if (live) {
            for (int c = 0; c < n_cand; ++c) {
                const R r2 = s_sq[c][0][u] + s_sq[c][1][v] + s_sq[c][2][w];  // dxdx + dydy + dzdz (:718)
                if (r2 < P.radius_sq) acc += s_vol[c] * ss_kernel_evaluate<R>(ss_sqrt(r2), P.h, P.sigma);  // :719-726
            }
        }
        __syncthreads();
    }
    if (live) G[((size_t)gi * P.np[1] + gj) * P.np[2] + gk] = acc;
}
template <class R>
void ssg_launch_levelset(const SSGlobT<R>& P, const R* xyz, const R* rho, const int* boxes, R* G, hipStream_t st) {
    const unsigned tiles = (unsigned)(((P.np[0] + SS_GTILE - 1) / SS_GTILE) * ((P.np[1] + SS_GTILE - 1) / SS_GTILE) * ((P.np[2] + SS_GTILE - 1) / SS_GTILE));
    if (!tiles) return;
    hipLaunchKernelGGL(k_g_levelset<R>, dim3(tiles), dim3(512), 0, st, P, xyz, rho, boxes, G);
}

// =====================================================================================================
// marching cubes of the global strategy (narrow_band_extraction.rs, triangulation.rs)
// =====================================================================================================
// An edge carries an iso-surface vertex iff one endpoint has a value >= t (it is in the map and not skipped,
// :69-71) and the other one a value < t (missing = 0, :79-92).  emask bit a: the edge from this point in +a.
template <class R>
__global__ __launch_bounds__(256) void k_g_edge_masks(SSGlobT<R> P, const R* __restrict__ G, uint8_t* __restrict__ emask, uint32_t* __restrict__ vcount) {
    const size_t npts = (size_t)P.np[0] * P.np[1] * P.np[2];
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= npts) return;
    const int k = (int)(f % (size_t)P.np[2]), j = (int)((f / (size_t)P.np[2]) % (size_t)P.np[1]), i = (int)(f / ((size_t)P.np[2] * P.np[1]));
    const int o[3] = {i, j, k};
    const size_t stride[3] = {(size_t)P.np[1] * P.np[2], (size_t)P.np[2], 1};
    const R vo = G[f];
    const bool o_low = vo < P.threshold;
    unsigned m = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (o[a] + 1 >= P.np[a]) continue;  // get_point_neighbor: None outside the grid (uniform_grid.rs:471-490)
        const bool q_low = G[f + stride[a]] < P.threshold;
        if (o_low != q_low) m |= 1u << a;
    }
    emask[f] = (uint8_t)m;
    vcount[f] = (uint32_t)__popc(m);
}
template <class R>
void ssg_launch_edge_masks(const SSGlobT<R>& P, const R* G, uint8_t* emask, uint32_t* vcount, hipStream_t st) {
    const size_t npts = (size_t)P.np[0] * P.np[1] * P.np[2];
    if (!npts) return;
    hipLaunchKernelGGL(k_g_edge_masks<R>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, P, G, emask, vcount);
}

// crossing bits of the 12 edges of cell (i,j,k) and its marching-cubes case (corner flags :115-127, 161-176)
template <class R>
__device__ inline int ssg_cell_case(const SSGlobT<R>& P, const R* __restrict__ G, const uint8_t* __restrict__ emask, int i, int j, int k, unsigned* edges_out) {
    unsigned edges = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const int oc = g_edge[e][0], a = g_edge[e][1];
        const size_t fo = ((size_t)(i + g_corner[oc][0]) * P.np[1] + (j + g_corner[oc][1])) * P.np[2] + (k + g_corner[oc][2]);
        if ((emask[fo] >> a) & 1u) edges |= 1u << e;
    }
    *edges_out = edges;
    if (!edges) return 0;
    int case_index = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const R val = G[((size_t)(i + g_corner[c][0]) * P.np[1] + (j + g_corner[c][1])) * P.np[2] + (k + g_corner[c][2])];
        bool above = val > P.threshold;
        if (!above && !(val < P.threshold)) {
            // value == t: flagged Above only through a crossing edge of this cell that touches the corner (:115-127)
            for (int e = 0; e < 12; ++e) {
                if (!((edges >> e) & 1u)) continue;
                const int oc = g_edge[e][0], a = g_edge[e][1];
                int tc[3] = {g_corner[oc][0], g_corner[oc][1], g_corner[oc][2]};
                const bool is_o = tc[0] == g_corner[c][0] && tc[1] == g_corner[c][1] && tc[2] == g_corner[c][2];
                tc[a] += 1;
                const bool is_t = tc[0] == g_corner[c][0] && tc[1] == g_corner[c][1] && tc[2] == g_corner[c][2];
                if (is_o || is_t) above = true;
            }
        }
        case_index |= (above ? 1 : 0) << c;
    }
    return case_index;
}

template <class R>
__global__ __launch_bounds__(256) void k_g_cell_count(SSGlobT<R> P, const R* __restrict__ G, const uint8_t* __restrict__ emask, uint32_t* __restrict__ tcount,
                                                      uint32_t* __restrict__ err) {
    const size_t ncell = (size_t)P.nc[0] * P.nc[1] * P.nc[2];
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ncell) return;
    const int k = (int)(f % (size_t)P.nc[2]), j = (int)((f / (size_t)P.nc[2]) % (size_t)P.nc[1]), i = (int)(f / ((size_t)P.nc[2] * P.nc[1]));
    unsigned edges = 0;
    const int case_index = ssg_cell_case(P, G, emask, i, j, k, &edges);
    uint32_t nt = 0;
    if (edges) {
        for (int t = 0; t < 5 && g_mc_table[case_index][3 * t] >= 0; ++t) {
            ++nt;
            for (int q = 0; q < 3; ++q)
                if (!((edges >> g_mc_table[case_index][3 * t + q]) & 1u)) atomicOr(err, 2u);  // "Missing iso surface vertex" (triangulation.rs:62-95)
        }
    }
    tcount[f] = nt;
}
template <class R>
void ssg_launch_cell_count(const SSGlobT<R>& P, const R* G, const uint8_t* emask, uint32_t* tcount, uint32_t* err, hipStream_t st) {
    const size_t ncell = (size_t)P.nc[0] * P.nc[1] * P.nc[2];
    if (!ncell) return;
    hipLaunchKernelGGL(k_g_cell_count<R>, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, P, G, emask, tcount, err);
}

template <class R>
__global__ __launch_bounds__(256) void k_g_emit_vertices(SSGlobT<R> P, const R* __restrict__ G, const uint8_t* __restrict__ emask, const uint32_t* __restrict__ vbase,
                                                         R* __restrict__ vertices, unsigned long long* __restrict__ vkeys) {
    const size_t npts = (size_t)P.np[0] * P.np[1] * P.np[2];
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= npts) return;
    const unsigned m = emask[f];
    if (!m) return;
    const int k = (int)(f % (size_t)P.np[2]), j = (int)((f / (size_t)P.np[2]) % (size_t)P.np[1]), i = (int)(f / ((size_t)P.np[2] * P.np[1]));
    const int o[3] = {i, j, k};
    const size_t stride[3] = {(size_t)P.np[1] * P.np[2], (size_t)P.np[2], 1};
    const R vo = G[f];
    uint32_t vid = vbase[f];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!((m >> a) & 1u)) continue;
        const R vq = G[f + stride[a]];
        // the edge is visited from its endpoint with value >= t ("point") towards the one < t ("neighbor") (:69-92)
        const bool from_o = !(vo < P.threshold);
        const R pv = from_o ? vo : vq, nv = from_o ? vq : vo;
        const R alpha = (P.threshold - pv) / (nv - pv);  // :95
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int po = o[d] + ((!from_o && d == a) ? 1 : 0), no = o[d] + ((from_o && d == a) ? 1 : 0);
            const R pc = P.gmin[d] + (R)po * P.cs, nc = P.gmin[d] + (R)no * P.cs;  // point_coordinates (uniform_grid.rs:418-437)
            vertices[3 * (size_t)vid + d] = pc * (R(1.0) - alpha) + nc * alpha;    // :96-99
        }
        vkeys[vid] = (unsigned long long)f * 3ull + (unsigned long long)a;
        ++vid;
    }
}
template <class R>
void ssg_launch_emit_vertices(const SSGlobT<R>& P, const R* G, const uint8_t* emask, const uint32_t* vbase, R* vertices, unsigned long long* vkeys, hipStream_t st) {
    const size_t npts = (size_t)P.np[0] * P.np[1] * P.np[2];
    if (!npts) return;
    hipLaunchKernelGGL(k_g_emit_vertices<R>, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, st, P, G, emask, vbase, vertices, vkeys);
}

template <class R>
__global__ __launch_bounds__(256) void k_g_emit_triangles(SSGlobT<R> P, const R* __restrict__ G, const uint8_t* __restrict__ emask, const uint32_t* __restrict__ vbase,
                                                          const uint32_t* __restrict__ tcount, const uint32_t* __restrict__ tbase, uint32_t* __restrict__ triangles) {
    const size_t ncell = (size_t)P.nc[0] * P.nc[1] * P.nc[2];
    const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ncell) return;
    const uint32_t nt = tcount[f];
    if (!nt) return;
    const int k = (int)(f % (size_t)P.nc[2]), j = (int)((f / (size_t)P.nc[2]) % (size_t)P.nc[1]), i = (int)(f / ((size_t)P.nc[2] * P.nc[1]));
    unsigned edges = 0;
    const int case_index = ssg_cell_case(P, G, emask, i, j, k, &edges);
    size_t o = 3 * (size_t)tbase[f];
    for (uint32_t t = 0; t < nt; ++t)
        for (int q = 0; q < 3; ++q) {
            const int e = g_mc_table[case_index][3 * t + q];
            const int oc = g_edge[e][0], a = g_edge[e][1];
            const size_t fo = ((size_t)(i + g_corner[oc][0]) * P.np[1] + (j + g_corner[oc][1])) * P.np[2] + (k + g_corner[oc][2]);
            const unsigned m = emask[fo];
            triangles[o++] = vbase[fo] + (uint32_t)__popc(m & ((1u << a) - 1u));
        }
}
template <class R>
void ssg_launch_emit_triangles(const SSGlobT<R>& P, const R* G, const uint8_t* emask, const uint32_t* vbase, const uint32_t* tcount, const uint32_t* tbase,
                               uint32_t* triangles, hipStream_t st) {
    const size_t ncell = (size_t)P.nc[0] * P.nc[1] * P.nc[2];
    if (!ncell) return;
    hipLaunchKernelGGL(k_g_emit_triangles<R>, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, st, P, G, emask, vbase, tcount, tbase, triangles);
}

// ---- explicit instantiations ----
#define SSG_INSTANTIATE(R)                                                                                                                             \
    template void ssg_launch_cell_keys<R>(const SSGlobT<R>&, const R*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, hipStream_t);                      \
    template void ssg_launch_density<R>(const SSGlobT<R>&, const R*, const uint32_t*, const uint32_t*, R*, int, uint32_t*, const unsigned long long*, \
                                        uint32_t*, hipStream_t);                                                                                     \
    template void ssg_launch_chunk_boxes<R>(const SSGlobT<R>&, const R*, int*, hipStream_t);                                                          \
    template void ssg_launch_levelset<R>(const SSGlobT<R>&, const R*, const R*, const int*, R*, hipStream_t);                                         \
    template void ssg_launch_edge_masks<R>(const SSGlobT<R>&, const R*, uint8_t*, uint32_t*, hipStream_t);                                            \
    template void ssg_launch_cell_count<R>(const SSGlobT<R>&, const R*, const uint8_t*, uint32_t*, uint32_t*, hipStream_t);                           \
    template void ssg_launch_emit_vertices<R>(const SSGlobT<R>&, const R*, const uint8_t*, const uint32_t*, R*, unsigned long long*, hipStream_t);    \
    template void ssg_launch_emit_triangles<R>(const SSGlobT<R>&, const R*, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*,        \
                                               uint32_t*, hipStream_t);
SSG_INSTANTIATE(float)
SSG_INSTANTIATE(double)
