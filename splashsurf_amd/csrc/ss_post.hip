// ss_post.hip -- on-device post-processing that consumes the mesh right after the reconstruction:
// vertex connectivity, area-weighted vertex normals, weighted Laplacian smoothing, normal smoothing, SPH interpolation
// of normals / particle attributes, and the smoothing weights of the reference's CLI recipe.  Kernels + C ABI
// (include/splashsurf_hip.h, section "post-processing").  Citations are relative to /root/reference/.
//
// Every array argument may be a host or an HBM pointer; HBM pointers are used in place (no D2H/H2D round trip
// between the reconstruction and these stages), host pointers are staged through temporary device buffers.
//
// Exactness: the stages whose reference functions are deterministic for a given mesh / connectivity (connectivity
// order, Laplacian smoothing incl. the reference's stale self term, normal smoothing, weighted neighbour counts,
// smooth-step weights) are reproduced bit for bit; vertex normals follow the reference's SEQUENTIAL function
// (mesh.rs:782-796; its parallel twin merges thread-local partial sums); the SPH sums run over the 27 cells of a
// uniform grid in lexicographic cell order (the reference: traversal order of an rstar R-tree) -- see oracle/splash_post.c.
#include <hip/hip_runtime.h>

#include <string.h>

#include <cmath>
#include <limits>
#include <string>

#include "ss_host.h"
#include "ss_kernels.h"

namespace {

// ---- host/device argument staging ----
// scratch from the context's grow-only pool (no hipMalloc/hipFree per call once the pool is warm)
inline ss_status pool_alloc(ss_context* ctx, size_t bytes, void** out) {
    if (ctx->post_pool_next >= (int)(sizeof(ctx->post_pool) / sizeof(ctx->post_pool[0]))) return fail(ctx, SS_ERR_UNKNOWN, "post-processing scratch pool exhausted");
    DevBuf& b = ctx->post_pool[ctx->post_pool_next++];
    SS_HIP(ctx, b.reserve(bytes ? bytes : 16));
    *out = b.p;
    return SS_OK;
}

template <class T>
struct DevArg {
    T* d = nullptr;        // device pointer to use
    void* host = nullptr;  // caller's host pointer (nullptr: the caller passed a device pointer)
    size_t n = 0;
    // mode: 0 input, 1 output, 2 in/out
    ss_status init(ss_context* ctx, const T* p, size_t count, int mode) {
        n = count;
        if (count == 0) return SS_OK;
        if (!p) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "null array argument");
        if (is_device_pointer(p)) {
            d = const_cast<T*>(p);
            return SS_OK;
        }
        host = const_cast<void*>(static_cast<const void*>(p));
        void* raw = nullptr;
        ss_status s = pool_alloc(ctx, count * sizeof(T), &raw);
        if (s != SS_OK) return s;
        d = static_cast<T*>(raw);
        if (mode != 1) SS_HIP(ctx, hipMemcpyAsync(raw, p, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return SS_OK;
    }
    ss_status finish(ss_context* ctx, size_t count = (size_t)-1) {
        if (host && d) {
            const size_t c = count == (size_t)-1 ? n : count;
            if (c) SS_HIP(ctx, hipMemcpyAsync(host, static_cast<const void*>(d), c * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
        }
        return SS_OK;
    }
};

struct TmpBuf {
    void* p = nullptr;
    ss_status alloc(ss_context* ctx, size_t bytes) { return pool_alloc(ctx, bytes, &p); }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

// the library's own primitives (ss_prims.h): a single-dispatch chained scan and the 8-bit LSD pair sort
template <class T>
struct PostArrayIn {
    const T* p;
    __device__ T operator()(uint32_t i) const { return p[i]; }
};
template <class T>
struct PostExclOut {
    T* p;
    __device__ void operator()(uint32_t i, T, T excl) const { p[i] = excl; }
};
template <class T>
ss_status scan_exclusive(ss_context* ctx, const T* in, T* out, size_t n) {
    if (n > SS_SCAN_MAX_N) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^32 - 8193 entries in one prefix sum");
    const size_t words = ss_scan_state_words(n);
    SS_HIP(ctx, ctx->temp.reserve(words * 4));
    SS_HIP(ctx, hipMemsetAsync(ctx->temp.p, 0, words * 4, ctx->stream));
    ss_chained_scan<T, SSOpPlus>(PostArrayIn<T>{in}, PostExclOut<T>{out}, (uint32_t)n, ctx->temp.as<uint32_t>(), (T*)nullptr, SSMailSlot{}, ctx->stream);
    return SS_OK;
}

// stable sort of (key, value) pairs by key <= max_key into keys_out / vals_out; the inputs are left untouched (they may be the caller's arrays): the sort's
// ping-pong buffers are the outputs and two scratch arrays, arranged so that the last pass writes the outputs
ss_status sort_pairs_u32(ss_context* ctx, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n, uint64_t max_key) {
    if (n == 0) return SS_OK;
    if (n >= ((size_t)1 << 30)) return fail(ctx, SS_ERR_UNSUPPORTED, "more than 2^30 - 1 entries to sort in one post-processing call are not supported by this build");
    unsigned bits = 1;
    while (bits < 32 && ((uint64_t)1 << bits) <= max_key) ++bits;
    const bool odd = (((bits + 7u) / 8u) & 1u) != 0u;
    const size_t work_words = ss_radix_sort_work_words((uint32_t)n, bits);
    SS_HIP(ctx, ctx->temp.reserve((2 * (n + 16) + work_words) * 4 + 64));
    uint32_t* sk = ctx->temp.as<uint32_t>();
    uint32_t* sv = sk + (n + 16);
    uint32_t* work = sv + (n + 16);
    // an odd number of passes ends in buffer 1, an even number in buffer 0: the outputs sit where the result lands, the input copies go into buffer 0
    uint32_t* keys[2] = {odd ? sk : keys_out, odd ? keys_out : sk};
    uint32_t* vals[2] = {odd ? sv : vals_out, odd ? vals_out : sv};
    SS_HIP(ctx, hipMemcpyAsync(keys[0], keys_in, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    SS_HIP(ctx, hipMemcpyAsync(vals[0], vals_in, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    const int r = ss_radix_sort_pairs(keys, vals, (uint32_t)n, bits, false, work, false, ctx->stream);
    if (keys[r] != keys_out || vals[r] != vals_out) return fail(ctx, SS_ERR_UNKNOWN, "internal error: sort result in an unexpected buffer");
    return SS_OK;
}

// =====================================================================================================
// vertex -> incident (triangle, slot) entries in ascending triangle order
// =====================================================================================================
__global__ __launch_bounds__(256) void k_p_incidence_keys(const uint32_t* __restrict__ tris, size_t n3, uint32_t nv, uint32_t* __restrict__ vals, uint32_t* __restrict__ count,
                                                          uint32_t* __restrict__ err) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n3) return;
    vals[e] = (uint32_t)e;
    const uint32_t v = tris[e];
    if (v >= nv) {
        atomicOr(err, 1u);
        return;
    }
    atomicAdd(&count[v], 1u);
}

struct Incidence {
    TmpBuf start, entries, keys_sorted, vals, count;
};

ss_status build_incidence(ss_context* ctx, const uint32_t* d_tris, uint64_t nt, uint64_t nv, Incidence* inc) {
    const size_t n3 = (size_t)nt * 3;
    ss_status s;
    if ((s = inc->start.alloc(ctx, (nv + 1) * 4)) != SS_OK) return s;
    if ((s = inc->count.alloc(ctx, (nv + 1) * 4)) != SS_OK) return s;
    if ((s = inc->entries.alloc(ctx, n3 * 4)) != SS_OK) return s;
    if ((s = inc->keys_sorted.alloc(ctx, n3 * 4)) != SS_OK) return s;
    if ((s = inc->vals.alloc(ctx, n3 * 4)) != SS_OK) return s;
    SS_HIP(ctx, ctx->counter.reserve(64));
    SS_HIP(ctx, hipMemsetAsync(ctx->counter.p, 0, 64, ctx->stream));
    SS_HIP(ctx, hipMemsetAsync(inc->count.p, 0, (nv + 1) * 4, ctx->stream));
    if (n3) {
        hipLaunchKernelGGL(k_p_incidence_keys, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, ctx->stream, d_tris, n3, (uint32_t)nv, inc->vals.as<uint32_t>(),
                           inc->count.as<uint32_t>(), ctx->counter.as<uint32_t>());
        uint32_t herr = 0;
        SS_HIP(ctx, hipMemcpyAsync(&herr, ctx->counter.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (herr) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "triangle refers to a vertex index >= n_vertices");
        // stable sort by vertex id: entries of a vertex stay in ascending (triangle, slot) order
        if ((s = sort_pairs_u32(ctx, d_tris, inc->keys_sorted.as<uint32_t>(), inc->vals.as<uint32_t>(), inc->entries.as<uint32_t>(), n3, nv)) != SS_OK) return s;
    }
    return scan_exclusive<uint32_t>(ctx, inc->count.as<uint32_t>(), inc->start.as<uint32_t>(), (size_t)nv + 1);
}

// =====================================================================================================
// vertex_vertex_connectivity (splashsurf_lib/src/mesh.rs:290-306): neighbours in first-occurrence order
// =====================================================================================================
template <int MODE>
__global__ __launch_bounds__(256) void k_p_connectivity(const uint32_t* __restrict__ tris, uint32_t nv, const uint32_t* __restrict__ inc_start,
                                                        const uint32_t* __restrict__ inc, unsigned long long* __restrict__ count,
                                                        const unsigned long long* __restrict__ row_ptr, uint32_t* __restrict__ nbrs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const uint32_t b = inc_start[i], e = inc_start[i + 1];
    unsigned long long n = 0;
    unsigned long long wr = (MODE == 1) ? row_ptr[i] : 0ull;
    for (uint32_t a = b; a < e; ++a) {
        const uint32_t t = inc[a] / 3u;
        for (int s = 0; s < 3; ++s) {
            const uint32_t j = tris[3 * (size_t)t + s];
            if (j == i) continue;
            bool seen = false;
            for (int s2 = 0; s2 < s && !seen; ++s2) seen = tris[3 * (size_t)t + s2] == j;
            for (uint32_t a2 = b; a2 < a && !seen; ++a2) {
                const uint32_t t2 = inc[a2] / 3u;
                if (t2 == t) continue;  // same triangle listed twice (degenerate): nothing new
                seen = tris[3 * (size_t)t2] == j || tris[3 * (size_t)t2 + 1] == j || tris[3 * (size_t)t2 + 2] == j;
            }
            if (seen) continue;
            if (MODE == 1) nbrs[wr++] = j;
            ++n;
        }
    }
    if (MODE == 0) count[i] = n;
}

// =====================================================================================================
// vertex normals (mesh.rs:782-796, 868-886): area-weighted sum in triangle order, then normalisation
// =====================================================================================================
template <class R>
__global__ __launch_bounds__(256) void k_p_vertex_normals(const R* __restrict__ v, const uint32_t* __restrict__ tris, uint32_t nv, const uint32_t* __restrict__ inc_start,
                                                          const uint32_t* __restrict__ inc, R* __restrict__ normals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    R acc[3] = {R(0.0), R(0.0), R(0.0)};
    for (uint32_t a = inc_start[i]; a < inc_start[i + 1]; ++a) {
        const uint32_t t = inc[a] / 3u;
        const R* v0 = v + 3 * (size_t)tris[3 * (size_t)t];
        const R* v1 = v + 3 * (size_t)tris[3 * (size_t)t + 1];
        const R* v2 = v + 3 * (size_t)tris[3 * (size_t)t + 2];
        const R ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2];
        const R bx = v2[0] - v1[0], by = v2[1] - v1[1], bz = v2[2] - v1[2];
        acc[0] += ay * bz - az * by;  // nalgebra cross
        acc[1] += az * bx - ax * bz;
        acc[2] += ax * by - ay * bx;
    }
    const R norm = ss_sqrt(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2]);
    normals[3 * (size_t)i] = acc[0] / norm;
    normals[3 * (size_t)i + 1] = acc[1] / norm;
    normals[3 * (size_t)i + 2] = acc[2] / norm;
}

// =====================================================================================================
// par_laplacian_smoothing_inplace (postprocessing.rs:17-52).  `cur` holds the values from TWO iterations ago when
// it is overwritten (the reference swaps its two buffers and blends into the stale one, :31-49); reproduced as is.
// =====================================================================================================
template <class R>
__global__ __launch_bounds__(256) void k_p_smooth_step(R* __restrict__ cur, const R* __restrict__ old, uint32_t nv, const unsigned long long* __restrict__ row_ptr,
                                                       const uint32_t* __restrict__ nbrs, R beta, const R* __restrict__ weights) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const R beta_eff = beta * (weights ? weights[i] : R(1.0));
    R sum[3] = {R(0.0), R(0.0), R(0.0)};
    const unsigned long long b = row_ptr[i], e = row_ptr[i + 1];
    for (unsigned long long q = b; q < e; ++q) {
        const R* o = old + 3 * (size_t)nbrs[q];
        sum[0] += o[0];
        sum[1] += o[1];
        sum[2] += o[2];
    }
    if (e > b) {
        const R n = (R)(double)(e - b);
        sum[0] /= n;
        sum[1] /= n;
        sum[2] /= n;
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) cur[3 * (size_t)i + d] = cur[3 * (size_t)i + d] * (R(1.0) - beta_eff) + sum[d] * beta_eff;
}

// par_laplacian_smoothing_normals_inplace (postprocessing.rs:55-96)
template <class R>
__global__ __launch_bounds__(256) void k_p_smooth_normals_step(R* __restrict__ out, const R* __restrict__ old, uint32_t nv, const unsigned long long* __restrict__ row_ptr,
                                                               const uint32_t* __restrict__ nbrs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    R s[3] = {R(0.0), R(0.0), R(0.0)};
    for (unsigned long long q = row_ptr[i]; q < row_ptr[i + 1]; ++q) {
        const R* o = old + 3 * (size_t)nbrs[q];
        s[0] += o[0];
        s[1] += o[1];
        s[2] += o[2];
    }
    const R norm = ss_sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    out[3 * (size_t)i] = s[0] / norm;
    out[3 * (size_t)i + 1] = s[1] / norm;
    out[3 * (size_t)i + 2] = s[2] / norm;
}

// splashsurf/src/reconstruct.rs:1189-1204
template <class R>
__global__ __launch_bounds__(256) void k_p_weighted_counts(const R* __restrict__ xyz, uint32_t n, const unsigned long long* __restrict__ nb_ptr,
                                                           const uint32_t* __restrict__ nb_idx, R squared_r, R* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const R px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    R acc = R(0.0);
    for (unsigned long long q = nb_ptr[i]; q < nb_ptr[i + 1]; ++q) {
        const R* pj = xyz + 3 * (size_t)nb_idx[q];
        const R dx = px - pj[0], dy = py - pj[1], dz = pz - pj[2];
        const R dist = dx * dx + dy * dy + dz * dz;
        R x = dist / squared_r;
        x = x < R(0.0) ? R(0.0) : (x > R(1.0) ? R(1.0) : x);
        acc = acc + (R(1.0) - x);
    }
    out[i] = acc;
}

// splashsurf/src/reconstruct.rs:1219-1232 (offset 0): clamp, normalise, smooth-step with powi by repeated squaring
template <class R>
__global__ __launch_bounds__(256) void k_p_smoothing_weights(const R* __restrict__ wnn, size_t n, R normalization, R* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    R v = wnn[i] - R(0.0);
    v = v > R(0.0) ? v : R(0.0);
    R x = v / (normalization - R(0.0));
    x = x < R(1.0) ? x : R(1.0);
    const R x2 = x * x, x4 = x2 * x2;
    const R x5 = x * x4, x3 = x * x2;
    out[i] = x5 * R(6.0) - x4 * R(15.0) + x3 * R(10.0);
}

// =====================================================================================================
// SPH interpolation (splashsurf_lib/src/sph_interpolation.rs) over a uniform cell grid of edge h
// =====================================================================================================
template <class R>
struct SphGrid {
    R origin[3];
    R h;
    int nc[3];
};

template <class R>
__global__ __launch_bounds__(256) void k_p_sph_keys(SphGrid<R> g, const R* __restrict__ xyz, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                    uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = (int)ss_floor((xyz[3 * (size_t)i + d] - g.origin[d]) / g.h);
    const uint32_t key = (uint32_t)((c[0] * g.nc[1] + c[1]) * g.nc[2] + c[2]);
    keys[i] = key;
    vals[i] = i;
    atomicAdd(&count[key], 1u);
}

template <class R>
__device__ inline R p_cubic_function_dq(R q) {  // kernel.rs:84-94
    const R pi = R(3.14159265358979323846);
    if (q < R(1.0)) return (R(3.0) / (R(4.0) * pi)) * (R(-4.0) * q + R(3.0) * q * q);
    if (q < R(2.0)) {
        const R x = R(2.0) - q;
        return -(R(3.0) / (R(4.0) * pi)) * x * x;
    }
    return R(0.0);
}

// interpolate_quantity_inplace (sph_interpolation.rs:205-259), DIM components per particle
template <class R, int DIM>
__global__ __launch_bounds__(256) void k_p_sph_interpolate(SphGrid<R> g, const R* __restrict__ xyz, const R* __restrict__ rho, const uint32_t* __restrict__ cell_start,
                                                           const uint32_t* __restrict__ items, R rest_mass, const R* __restrict__ values,
                                                           const R* __restrict__ points, size_t n_points, R enable, R* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const R x[3] = {points[3 * p], points[3 * p + 1], points[3 * p + 2]};
    const R squared_support = g.h * g.h;
    const R sigma = R(8.0) / (g.h * g.h * g.h);
    R acc[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) acc[k] = R(0.0);
    R correction = R(0.0);
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const R f = ss_floor((x[d] - g.origin[d]) / g.h);
        c[d] = (f < R(-2.0)) ? -2 : ((f > (R)(g.nc[d] + 1)) ? g.nc[d] + 1 : (int)f);
    }
    for (int cx = c[0] - 1; cx <= c[0] + 1; ++cx)
        for (int cy = c[1] - 1; cy <= c[1] + 1; ++cy)
            for (int cz = c[2] - 1; cz <= c[2] + 1; ++cz) {
                if (cx < 0 || cy < 0 || cz < 0 || cx >= g.nc[0] || cy >= g.nc[1] || cz >= g.nc[2]) continue;
                const uint32_t f = (uint32_t)((cx * g.nc[1] + cy) * g.nc[2] + cz);
                for (uint32_t q = cell_start[f]; q < cell_start[f + 1]; ++q) {
                    const uint32_t j = items[q];
                    const R dx = xyz[3 * (size_t)j] - x[0], dy = xyz[3 * (size_t)j + 1] - x[1], dz = xyz[3 * (size_t)j + 2] - x[2];
                    const R d2 = dx * dx + dy * dy + dz * dz;
                    if (!(d2 <= squared_support)) continue;
                    const R vol = rest_mass / rho[j];
                    const R r = ss_sqrt(d2);
                    const R w = ss_kernel_evaluate<R>(r, g.h, sigma);
                    const R vw = vol * w;
#pragma unroll
                    for (int k = 0; k < DIM; ++k) acc[k] += values[(size_t)DIM * j + k] * vw;
                    correction += vw;
                }
            }
    const R factor = enable * (R(1.0) / correction) + (R(1.0) - enable);
#pragma unroll
    for (int k = 0; k < DIM; ++k) out[(size_t)DIM * p + k] = acc[k] * factor;
}

// interpolate_normals_inplace (sph_interpolation.rs:72-113)
template <class R>
__global__ __launch_bounds__(256) void k_p_sph_normals(SphGrid<R> g, const R* __restrict__ xyz, const R* __restrict__ rho, const uint32_t* __restrict__ cell_start,
                                                       const uint32_t* __restrict__ items, R rest_mass, const R* __restrict__ points, size_t n_points,
                                                       R* __restrict__ out) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const R x[3] = {points[3 * p], points[3 * p + 1], points[3 * p + 2]};
    const R squared_support = g.h * g.h;
    const R sigma = R(8.0) / (g.h * g.h * g.h);
    R grad[3] = {R(0.0), R(0.0), R(0.0)};
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const R f = ss_floor((x[d] - g.origin[d]) / g.h);
        c[d] = (f < R(-2.0)) ? -2 : ((f > (R)(g.nc[d] + 1)) ? g.nc[d] + 1 : (int)f);
    }
    for (int cx = c[0] - 1; cx <= c[0] + 1; ++cx)
        for (int cy = c[1] - 1; cy <= c[1] + 1; ++cy)
            for (int cz = c[2] - 1; cz <= c[2] + 1; ++cz) {
                if (cx < 0 || cy < 0 || cz < 0 || cx >= g.nc[0] || cy >= g.nc[1] || cz >= g.nc[2]) continue;
                const uint32_t f = (uint32_t)((cx * g.nc[1] + cy) * g.nc[2] + cz);
                for (uint32_t q = cell_start[f]; q < cell_start[f + 1]; ++q) {
                    const uint32_t j = items[q];
                    const R dx[3] = {xyz[3 * (size_t)j] - x[0], xyz[3 * (size_t)j + 1] - x[1], xyz[3 * (size_t)j + 2] - x[2]};
                    const R d2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
                    if (!(d2 <= squared_support)) continue;
                    const R vol = rest_mass / rho[j];
                    const R r = ss_sqrt(d2);
                    const R q_ = (r + r) / g.h;
                    const R gnorm = sigma * p_cubic_function_dq<R>(q_) * ((R(1.0) + R(1.0)) / g.h);  // kernel.rs:132-139
#pragma unroll
                    for (int d = 0; d < 3; ++d) grad[d] += ((dx[d] / r) * gnorm) * vol;  // :103-104
                }
            }
    const R norm = ss_sqrt(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]);
#pragma unroll
    for (int d = 0; d < 3; ++d) out[3 * p + d] = grad[d] / norm;
}

template <class R>
struct SphIndex {
    SphGrid<R> g;
    TmpBuf cell_start, items, keys, keys_sorted, vals, count;
};

template <class R>
ss_status build_sph_index(ss_context* ctx, const R* d_xyz, uint64_t n, R h, SphIndex<R>* ix) {
    if (!(h > R(0.0))) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "compact support radius must be positive");
    if (n >= (1ull << 31)) return fail(ctx, SS_ERR_UNSUPPORTED, "too many particles");
    R mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    if (n > 0) {
        SS_HIP(ctx, ctx->aabb_partial.reserve(SS_AABB_PARTIAL_WORDS * sizeof(R)));
        SS_HIP(ctx, ctx->aabb_out.reserve(6 * sizeof(R)));
        ss_launch_aabb<R>(d_xyz, (uint32_t)n, ctx->aabb_partial.as<R>(), ctx->aabb_out.as<R>(), SSMailSlot{}, ctx->stream);
        R h6[6];
        SS_HIP(ctx, hipMemcpyAsync(h6, ctx->aabb_out.p, 6 * sizeof(R), hipMemcpyDeviceToHost, ctx->stream));
        SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int d = 0; d < 3; ++d) {
            mn[d] = h6[d];
            mx[d] = h6[3 + d];
        }
    }
    double ncell_d = 1.0;
    ix->g.h = h;
    for (int d = 0; d < 3; ++d) {
        ix->g.origin[d] = ss_floor(mn[d] / h) * h - h;
        const double nc = (double)ss_floor((mx[d] - ix->g.origin[d]) / h) + 2.0;
        if (!(nc >= 1.0 && nc < 2.0e9)) return fail(ctx, SS_ERR_UNSUPPORTED, "interpolation grid out of range");
        ix->g.nc[d] = (int)nc;
        ncell_d *= nc;
    }
    if (ncell_d >= 4.0e9) return fail(ctx, SS_ERR_UNSUPPORTED, "interpolation grid too large for the dense cell table of this build");
    const size_t ncell = (size_t)ncell_d;
    ss_status s;
    if ((s = ix->cell_start.alloc(ctx, (ncell + 1) * 4)) != SS_OK) return s;
    if ((s = ix->count.alloc(ctx, (ncell + 1) * 4)) != SS_OK) return s;
    if ((s = ix->items.alloc(ctx, n * 4)) != SS_OK) return s;
    if ((s = ix->keys.alloc(ctx, n * 4)) != SS_OK) return s;
    if ((s = ix->keys_sorted.alloc(ctx, n * 4)) != SS_OK) return s;
    if ((s = ix->vals.alloc(ctx, n * 4)) != SS_OK) return s;
    SS_HIP(ctx, hipMemsetAsync(ix->count.p, 0, (ncell + 1) * 4, ctx->stream));
    if (n) {
        hipLaunchKernelGGL(k_p_sph_keys<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ix->g, d_xyz, (uint32_t)n, ix->keys.template as<uint32_t>(),
                           ix->vals.template as<uint32_t>(), ix->count.template as<uint32_t>());
        if ((s = sort_pairs_u32(ctx, ix->keys.template as<uint32_t>(), ix->keys_sorted.template as<uint32_t>(), ix->vals.template as<uint32_t>(), ix->items.template as<uint32_t>(), n, ncell)) != SS_OK)
            return s;
    }
    return scan_exclusive<uint32_t>(ctx, ix->count.template as<uint32_t>(), ix->cell_start.template as<uint32_t>(), ncell + 1);
}

ss_status sync_and_check(ss_context* ctx) {
    SS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SS_ERR_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(e));
    return SS_OK;
}

#define SS_TRY(expr)                 \
    do {                             \
        ss_status _s = (expr);       \
        if (_s != SS_OK) return _s;  \
    } while (0)

ss_status begin_call(ss_context* c) {
    if (!c) return SS_ERR_INVALID_ARGUMENT;
    c->err.clear();
    c->err_detail = 0;
    SS_HIP(c, hipSetDevice(c->device));
    c->post_pool_next = 0;
    return SS_OK;
}

// ---- templated implementations of the ABI functions ----
template <class R>
ss_status vertex_normals_impl(ss_context* c, const R* vertices, uint64_t nv, const uint32_t* tris, uint64_t nt, R* normals) {
    SS_TRY(begin_call(c));
    if (nv >= (1ull << 32) || nt * 3 >= (1ull << 32)) return fail(c, SS_ERR_UNSUPPORTED, "mesh too large for 32-bit indices");
    DevArg<const R> v;
    DevArg<const uint32_t> t;
    DevArg<R> out;
    SS_TRY(v.init(c, vertices, (size_t)nv * 3, 0));
    SS_TRY(t.init(c, tris, (size_t)nt * 3, 0));
    SS_TRY(out.init(c, normals, (size_t)nv * 3, 1));
    Incidence inc;
    SS_TRY(build_incidence(c, t.d, nt, nv, &inc));
    if (nv)
        hipLaunchKernelGGL(k_p_vertex_normals<R>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, c->stream, v.d, t.d, (uint32_t)nv, inc.start.as<uint32_t>(),
                           inc.entries.as<uint32_t>(), out.d);
    SS_TRY(out.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status smoothing_impl(ss_context* c, R* vertices, uint64_t nv, const uint64_t* row_ptr, const uint32_t* nbrs, uint32_t iterations, R beta, const R* weights) {
    SS_TRY(begin_call(c));
    if (nv >= (1ull << 32)) return fail(c, SS_ERR_UNSUPPORTED, "mesh too large for 32-bit indices");
    DevArg<R> v;
    DevArg<const unsigned long long> row;
    DevArg<const uint32_t> nb;
    DevArg<const R> w;
    SS_TRY(v.init(c, vertices, (size_t)nv * 3, 2));
    SS_TRY(row.init(c, reinterpret_cast<const unsigned long long*>(row_ptr), (size_t)nv + 1, 0));
    uint64_t m = 0;
    if (nv) {
        SS_HIP(c, hipMemcpyAsync(&m, row.d + nv, 8, hipMemcpyDeviceToHost, c->stream));
        SS_HIP(c, hipStreamSynchronize(c->stream));
    }
    SS_TRY(nb.init(c, nbrs, (size_t)m, 0));
    if (weights) SS_TRY(w.init(c, weights, (size_t)nv, 0));
    TmpBuf buf;
    SS_TRY(buf.alloc(c, (size_t)nv * 3 * sizeof(R)));
    if (nv) SS_HIP(c, hipMemcpyAsync(buf.p, v.d, (size_t)nv * 3 * sizeof(R), hipMemcpyDeviceToDevice, c->stream));  // vertex_buffer = mesh.vertices.clone()
    R* cur = v.d;
    R* old = buf.as<R>();
    for (uint32_t it = 0; it < iterations && nv; ++it) {
        std::swap(cur, old);
        hipLaunchKernelGGL(k_p_smooth_step<R>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, c->stream, cur, old, (uint32_t)nv, row.d, nb.d, beta,
                           weights ? w.d : (const R*)nullptr);
    }
    if (cur != v.d && nv) SS_HIP(c, hipMemcpyAsync(v.d, cur, (size_t)nv * 3 * sizeof(R), hipMemcpyDeviceToDevice, c->stream));
    SS_TRY(v.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status smooth_normals_impl(ss_context* c, R* normals, uint64_t nv, const uint64_t* row_ptr, const uint32_t* nbrs, uint32_t iterations) {
    SS_TRY(begin_call(c));
    if (nv >= (1ull << 32)) return fail(c, SS_ERR_UNSUPPORTED, "mesh too large for 32-bit indices");
    DevArg<R> nrm;
    DevArg<const unsigned long long> row;
    DevArg<const uint32_t> nb;
    SS_TRY(nrm.init(c, normals, (size_t)nv * 3, 2));
    SS_TRY(row.init(c, reinterpret_cast<const unsigned long long*>(row_ptr), (size_t)nv + 1, 0));
    uint64_t m = 0;
    if (nv) {
        SS_HIP(c, hipMemcpyAsync(&m, row.d + nv, 8, hipMemcpyDeviceToHost, c->stream));
        SS_HIP(c, hipStreamSynchronize(c->stream));
    }
    SS_TRY(nb.init(c, nbrs, (size_t)m, 0));
    TmpBuf buf;
    SS_TRY(buf.alloc(c, (size_t)nv * 3 * sizeof(R)));
    R* old = buf.as<R>();
    R* smoothed = nrm.d;
    for (uint32_t it = 0; it < iterations && nv; ++it) {
        std::swap(old, smoothed);
        hipLaunchKernelGGL(k_p_smooth_normals_step<R>, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, c->stream, smoothed, old, (uint32_t)nv, row.d, nb.d);
    }
    if (smoothed != nrm.d && nv) SS_HIP(c, hipMemcpyAsync(nrm.d, smoothed, (size_t)nv * 3 * sizeof(R), hipMemcpyDeviceToDevice, c->stream));
    SS_TRY(nrm.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status weighted_counts_impl(ss_context* c, const R* xyz, uint64_t n, const uint64_t* nb_ptr, const uint32_t* nb_idx, R h, R* out) {
    SS_TRY(begin_call(c));
    if (n >= (1ull << 32)) return fail(c, SS_ERR_UNSUPPORTED, "too many particles");
    DevArg<const R> x;
    DevArg<const unsigned long long> row;
    DevArg<const uint32_t> idx;
    DevArg<R> o;
    SS_TRY(x.init(c, xyz, (size_t)n * 3, 0));
    SS_TRY(row.init(c, reinterpret_cast<const unsigned long long*>(nb_ptr), (size_t)n + 1, 0));
    uint64_t m = 0;
    if (n) {
        SS_HIP(c, hipMemcpyAsync(&m, row.d + n, 8, hipMemcpyDeviceToHost, c->stream));
        SS_HIP(c, hipStreamSynchronize(c->stream));
    }
    SS_TRY(idx.init(c, nb_idx, (size_t)m, 0));
    SS_TRY(o.init(c, out, (size_t)n, 1));
    if (n)
        hipLaunchKernelGGL(k_p_weighted_counts<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, x.d, (uint32_t)n, row.d, idx.d, h * h, o.d);
    SS_TRY(o.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status smoothing_weights_impl(ss_context* c, const R* wnn, uint64_t n, R normalization, R* out) {
    SS_TRY(begin_call(c));
    DevArg<const R> a;
    DevArg<R> o;
    SS_TRY(a.init(c, wnn, (size_t)n, 0));
    SS_TRY(o.init(c, out, (size_t)n, 1));
    if (n) hipLaunchKernelGGL(k_p_smoothing_weights<R>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, a.d, (size_t)n, normalization, o.d);
    SS_TRY(o.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status sph_interpolate_impl(ss_context* c, const R* xyz, const R* rho, uint64_t n, R rest_mass, R h, const R* values, int dim, const R* points, uint64_t np,
                               int first_order, R* out) {
    SS_TRY(begin_call(c));
    if (dim != 1 && dim != 3) return fail(c, SS_ERR_UNSUPPORTED, "interpolation of 1- and 3-component quantities only");
    DevArg<const R> x, r, v, p;
    DevArg<R> o;
    SS_TRY(x.init(c, xyz, (size_t)n * 3, 0));
    SS_TRY(r.init(c, rho, (size_t)n, 0));
    SS_TRY(v.init(c, values, (size_t)n * dim, 0));
    SS_TRY(p.init(c, points, (size_t)np * 3, 0));
    SS_TRY(o.init(c, out, (size_t)np * dim, 1));
    SphIndex<R> ix;
    SS_TRY(build_sph_index<R>(c, x.d, n, h, &ix));
    const R enable = first_order ? R(1.0) : R(0.0);
    if (np) {
        const dim3 g((unsigned)((np + 255) / 256)), b(256);
        if (dim == 1)
            hipLaunchKernelGGL((k_p_sph_interpolate<R, 1>), g, b, 0, c->stream, ix.g, x.d, r.d, ix.cell_start.template as<uint32_t>(), ix.items.template as<uint32_t>(), rest_mass, v.d, p.d,
                               (size_t)np, enable, o.d);
        else
            hipLaunchKernelGGL((k_p_sph_interpolate<R, 3>), g, b, 0, c->stream, ix.g, x.d, r.d, ix.cell_start.template as<uint32_t>(), ix.items.template as<uint32_t>(), rest_mass, v.d, p.d,
                               (size_t)np, enable, o.d);
    }
    SS_TRY(o.finish(c));
    return sync_and_check(c);
}

template <class R>
ss_status sph_normals_impl(ss_context* c, const R* xyz, const R* rho, uint64_t n, R rest_mass, R h, const R* points, uint64_t np, R* out) {
    SS_TRY(begin_call(c));
    DevArg<const R> x, r, p;
    DevArg<R> o;
    SS_TRY(x.init(c, xyz, (size_t)n * 3, 0));
    SS_TRY(r.init(c, rho, (size_t)n, 0));
    SS_TRY(p.init(c, points, (size_t)np * 3, 0));
    SS_TRY(o.init(c, out, (size_t)np * 3, 1));
    SphIndex<R> ix;
    SS_TRY(build_sph_index<R>(c, x.d, n, h, &ix));
    if (np)
        hipLaunchKernelGGL(k_p_sph_normals<R>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, c->stream, ix.g, x.d, r.d, ix.cell_start.template as<uint32_t>(),
                           ix.items.template as<uint32_t>(), rest_mass, p.d, (size_t)np, o.d);
    SS_TRY(o.finish(c));
    return sync_and_check(c);
}

}  // namespace

// =====================================================================================================
// C ABI (include/splashsurf_hip.h, "post-processing")
// =====================================================================================================
extern "C" {

ss_status ss_post_vertex_connectivity(ss_context* c, uint64_t n_vertices, const uint32_t* triangles, uint64_t n_triangles, uint64_t* row_ptr, uint32_t* neighbors,
                                      uint64_t neighbors_capacity, uint64_t* n_entries) {
    SS_TRY(begin_call(c));
    if (!n_entries) return fail(c, SS_ERR_INVALID_ARGUMENT, "n_entries is null");
    if (n_vertices >= (1ull << 32) || n_triangles * 3 >= (1ull << 32)) return fail(c, SS_ERR_UNSUPPORTED, "mesh too large for 32-bit indices");
    DevArg<const uint32_t> t;
    DevArg<unsigned long long> row;
    DevArg<uint32_t> nb;
    SS_TRY(t.init(c, triangles, (size_t)n_triangles * 3, 0));
    SS_TRY(row.init(c, reinterpret_cast<unsigned long long*>(row_ptr), (size_t)n_vertices + 1, 1));
    Incidence inc;
    SS_TRY(build_incidence(c, t.d, n_triangles, n_vertices, &inc));
    TmpBuf cnt;
    SS_TRY(cnt.alloc(c, ((size_t)n_vertices + 1) * 8));
    SS_HIP(c, hipMemsetAsync(cnt.p, 0, ((size_t)n_vertices + 1) * 8, c->stream));
    const dim3 g((unsigned)((n_vertices + 255) / 256)), b(256);
    if (n_vertices)
        hipLaunchKernelGGL(k_p_connectivity<0>, g, b, 0, c->stream, t.d, (uint32_t)n_vertices, inc.start.as<uint32_t>(), inc.entries.as<uint32_t>(),
                           cnt.as<unsigned long long>(), (const unsigned long long*)nullptr, (uint32_t*)nullptr);
    SS_TRY(scan_exclusive<unsigned long long>(c, cnt.as<unsigned long long>(), row.d, (size_t)n_vertices + 1));
    unsigned long long total = 0;
    SS_HIP(c, hipMemcpyAsync(&total, row.d + n_vertices, 8, hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    *n_entries = total;
    if (total > neighbors_capacity)
        return fail(c, SS_ERR_INVALID_ARGUMENT, "neighbors_capacity too small (6 * n_triangles always suffices); required size returned in n_entries");
    SS_TRY(nb.init(c, neighbors, (size_t)total, 1));
    if (n_vertices && total)
        hipLaunchKernelGGL(k_p_connectivity<1>, g, b, 0, c->stream, t.d, (uint32_t)n_vertices, inc.start.as<uint32_t>(), inc.entries.as<uint32_t>(),
                           (unsigned long long*)nullptr, row.d, nb.d);
    SS_TRY(row.finish(c));
    SS_TRY(nb.finish(c));
    return sync_and_check(c);
}

ss_status ss_post_vertex_normals_f32(ss_context* c, const float* vertices, uint64_t nv, const uint32_t* triangles, uint64_t nt, float* normals) {
    return vertex_normals_impl<float>(c, vertices, nv, triangles, nt, normals);
}
ss_status ss_post_vertex_normals_f64(ss_context* c, const double* vertices, uint64_t nv, const uint32_t* triangles, uint64_t nt, double* normals) {
    return vertex_normals_impl<double>(c, vertices, nv, triangles, nt, normals);
}

ss_status ss_post_laplacian_smoothing_f32(ss_context* c, float* vertices, uint64_t nv, const uint64_t* row_ptr, const uint32_t* neighbors, uint32_t iterations, float beta,
                                          const float* weights) {
    return smoothing_impl<float>(c, vertices, nv, row_ptr, neighbors, iterations, beta, weights);
}
ss_status ss_post_laplacian_smoothing_f64(ss_context* c, double* vertices, uint64_t nv, const uint64_t* row_ptr, const uint32_t* neighbors, uint32_t iterations, double beta,
                                          const double* weights) {
    return smoothing_impl<double>(c, vertices, nv, row_ptr, neighbors, iterations, beta, weights);
}

ss_status ss_post_smooth_normals_f32(ss_context* c, float* normals, uint64_t nv, const uint64_t* row_ptr, const uint32_t* neighbors, uint32_t iterations) {
    return smooth_normals_impl<float>(c, normals, nv, row_ptr, neighbors, iterations);
}
ss_status ss_post_smooth_normals_f64(ss_context* c, double* normals, uint64_t nv, const uint64_t* row_ptr, const uint32_t* neighbors, uint32_t iterations) {
    return smooth_normals_impl<double>(c, normals, nv, row_ptr, neighbors, iterations);
}

ss_status ss_post_weighted_neighbor_counts_f32(ss_context* c, const float* xyz, uint64_t n, const uint64_t* nb_row_ptr, const uint32_t* nb_indices, float h, float* out) {
    return weighted_counts_impl<float>(c, xyz, n, nb_row_ptr, nb_indices, h, out);
}
ss_status ss_post_weighted_neighbor_counts_f64(ss_context* c, const double* xyz, uint64_t n, const uint64_t* nb_row_ptr, const uint32_t* nb_indices, double h, double* out) {
    return weighted_counts_impl<double>(c, xyz, n, nb_row_ptr, nb_indices, h, out);
}

ss_status ss_post_smoothing_weights_f32(ss_context* c, const float* wnn, uint64_t n, float normalization, float* out) {
    return smoothing_weights_impl<float>(c, wnn, n, normalization, out);
}
ss_status ss_post_smoothing_weights_f64(ss_context* c, const double* wnn, uint64_t n, double normalization, double* out) {
    return smoothing_weights_impl<double>(c, wnn, n, normalization, out);
}

ss_status ss_post_sph_interpolate_f32(ss_context* c, const float* xyz, const float* rho, uint64_t n, float rest_mass, float h, const float* values, int32_t dim,
                                      const float* points, uint64_t n_points, int32_t first_order_correction, float* out) {
    return sph_interpolate_impl<float>(c, xyz, rho, n, rest_mass, h, values, dim, points, n_points, first_order_correction, out);
}
ss_status ss_post_sph_interpolate_f64(ss_context* c, const double* xyz, const double* rho, uint64_t n, double rest_mass, double h, const double* values, int32_t dim,
                                      const double* points, uint64_t n_points, int32_t first_order_correction, double* out) {
    return sph_interpolate_impl<double>(c, xyz, rho, n, rest_mass, h, values, dim, points, n_points, first_order_correction, out);
}

ss_status ss_post_sph_normals_f32(ss_context* c, const float* xyz, const float* rho, uint64_t n, float rest_mass, float h, const float* points, uint64_t n_points, float* out) {
    return sph_normals_impl<float>(c, xyz, rho, n, rest_mass, h, points, n_points, out);
}
ss_status ss_post_sph_normals_f64(ss_context* c, const double* xyz, const double* rho, uint64_t n, double rest_mass, double h, const double* points, uint64_t n_points,
                                  double* out) {
    return sph_normals_impl<double>(c, xyz, rho, n, rest_mass, h, points, n_points, out);
}

// device-resident views of the reconstruction for zero-copy post-processing
ss_status ss_result_device_particle_neighbors(const ss_result* r, const uint64_t** row_ptr, const uint32_t** neighbors, uint64_t* n_particles, uint64_t* n_entries) {
    if (!r || !r->valid || !row_ptr || !neighbors || !n_particles || !n_entries) return SS_ERR_INVALID_ARGUMENT;
    *n_particles = r->n_particles;
    if (!r->has_neighbors) {
        *row_ptr = nullptr;
        *neighbors = nullptr;
        *n_entries = 0;
        return SS_OK;
    }
    *row_ptr = reinterpret_cast<const uint64_t*>(r->nb_ptr.as<unsigned long long>());
    *neighbors = r->nb_idx.as<uint32_t>();
    *n_entries = r->n_neighbors;
    return SS_OK;
}

// copies of the reconstruction's arrays into caller buffers (host or HBM): the input of the post-processing stages,
// which modify their arrays in place while the ss_result keeps the raw mesh
static ss_status copy_out(ss_result* r, const void* src, void* dst, size_t bytes) {
    if (!r || !r->valid || !dst) return SS_ERR_INVALID_ARGUMENT;
    ss_context* c = r->ctx;
    if (!bytes) return SS_OK;
    SS_HIP(c, hipSetDevice(c->device));
    SS_HIP(c, hipMemcpyAsync(dst, src, bytes, is_device_pointer(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}
ss_status ss_result_copy_vertices(ss_result* r, void* dst) {
    if (!r) return SS_ERR_INVALID_ARGUMENT;
    return copy_out(r, r->vertices.p, dst, (size_t)r->n_vertices * 3 * (r->is_f64 ? 8 : 4));
}
ss_status ss_result_copy_triangles_u32(ss_result* r, uint32_t* dst) {
    if (!r) return SS_ERR_INVALID_ARGUMENT;
    return copy_out(r, r->tri32.p, dst, (size_t)r->n_triangles * 12);
}
ss_status ss_result_copy_vertex_keys(ss_result* r, uint64_t* dst) {
    if (!r) return SS_ERR_INVALID_ARGUMENT;
    return copy_out(r, r->vkeys.p, dst, (size_t)r->n_vertices * 8);
}
ss_status ss_result_copy_particle_densities(ss_result* r, void* dst) {
    if (!r) return SS_ERR_INVALID_ARGUMENT;
    return copy_out(r, r->rho.p, dst, (size_t)r->n_particles * (r->is_f64 ? 8 : 4));
}

}  // extern "C"
