/* splash_post.c -- CPU ORACLE (test infrastructure, NOT the product) for the post-processing stages that
 * consume the mesh right after the reconstruction.  Same rules as splash_oracle.c:
 * plain C, compiled with -ffp-contract=off for the two Real types (so_post_*, so64_post_*), every function
 * cites the reference file:line it restates (paths relative to /root/reference/).
 *
 * Pinning (tools/gen_goldens.py --post-only, tests/golden/post_*.npz):
 *   - vertex connectivity, weighted Laplacian smoothing, normal smoothing: the reference functions are
 *     deterministic for a given mesh/connectivity -> bit-identical to the wheel's outputs;
 *   - area-weighted vertex normals: the restatement follows the SEQUENTIAL function (mesh.rs:782-796); the wheel only
 *     exposes the parallel one (thread-local partial sums, mesh.rs:798-838) -> pinned within 1e-5;
 *   - SPH interpolation: the reference sums in the traversal order of an R-tree built by the third-party crate
 *     rstar 0.12 (Cargo.lock; sph_interpolation.rs:86-88, 312-316), which is not part of the reference tree and is
 *     not restated.  This oracle sums over the 27 cells of a uniform grid (cell size h, cells in lexicographic
 *     order, ascending particle index inside a cell) -- the same particles, another order -> pinned within 1e-5
 *     relative on values / 1e-4 on unit normals.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef SO_F64
typedef double real;
#define PFN(name) so64_post_##name
#define R_SQRT sqrt
#define R_FLOOR floor
#else
typedef float real;
#define PFN(name) so_post_##name
#define R_SQRT sqrtf
#define R_FLOOR floorf
#endif
#define RC(x) ((real)(x))

/* mesh.rs:290-306 (vertex_vertex_connectivity): neighbours in first-occurrence order while scanning the
 * triangles in order.  Returns CSR arrays allocated with malloc. */
int PFN(vertex_connectivity)(uint64_t n_vertices, const uint64_t *tris, uint64_t n_tris, uint64_t **row_ptr_out, uint32_t **nbrs_out) {
    uint32_t *cnt = (uint32_t *)calloc(n_vertices + 1, sizeof(uint32_t));
    uint32_t *cap = (uint32_t *)calloc(n_vertices + 1, sizeof(uint32_t));
    uint32_t **lists = (uint32_t **)calloc(n_vertices + 1, sizeof(uint32_t *));
    for (uint64_t t = 0; t < n_tris; ++t)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                const uint64_t i = tris[3 * t + a], j = tris[3 * t + b];
                if (i == j) continue;
                int found = 0;
                for (uint32_t q = 0; q < cnt[i]; ++q)
                    if (lists[i][q] == (uint32_t)j) {
                        found = 1;
                        break;
                    }
                if (found) continue;
                if (cnt[i] == cap[i]) {
                    cap[i] = cap[i] ? cap[i] * 2 : 8;
                    lists[i] = (uint32_t *)realloc(lists[i], sizeof(uint32_t) * cap[i]);
                }
                lists[i][cnt[i]++] = (uint32_t)j;
            }
    uint64_t *row = (uint64_t *)malloc(sizeof(uint64_t) * (n_vertices + 1));
    uint64_t run = 0;
    for (uint64_t i = 0; i < n_vertices; ++i) {
        row[i] = run;
        run += cnt[i];
    }
    row[n_vertices] = run;
    uint32_t *nb = (uint32_t *)malloc(sizeof(uint32_t) * (run ? run : 1));
    for (uint64_t i = 0; i < n_vertices; ++i) {
        memcpy(nb + row[i], lists[i], sizeof(uint32_t) * cnt[i]);
        free(lists[i]);
    }
    free(lists);
    free(cnt);
    free(cap);
    *row_ptr_out = row;
    *nbrs_out = nb;
    return 0;
}

void PFN(free)(void *p) { free(p); }

/* mesh.rs:782-796 + 868-886 (vertex_normals, sequential): area-weighted sum in triangle order, then normalisation */
void PFN(vertex_normals)(const real *v, uint64_t n_vertices, const uint64_t *tris, uint64_t n_tris, real *normals) {
    memset(normals, 0, sizeof(real) * 3 * n_vertices);
    for (uint64_t t = 0; t < n_tris; ++t) {
        const real *v0 = v + 3 * tris[3 * t], *v1 = v + 3 * tris[3 * t + 1], *v2 = v + 3 * tris[3 * t + 2];
        const real a[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
        const real b[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]};
        const real n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; /* nalgebra cross */
        for (int k = 0; k < 3; ++k)
            for (int d = 0; d < 3; ++d) normals[3 * tris[3 * t + k] + d] += n[d];
    }
    for (uint64_t i = 0; i < n_vertices; ++i) {
        real *n = normals + 3 * i;
        const real norm = R_SQRT(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]); /* mesh.rs:880 */
        for (int d = 0; d < 3; ++d) n[d] /= norm;
    }
}

/* postprocessing.rs:17-52 (par_laplacian_smoothing_inplace) */
void PFN(laplacian_smoothing)(real *vertices, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *nbrs, uint32_t iterations, real beta,
                              const real *weights) {
    real *buf = (real *)malloc(sizeof(real) * 3 * (n_vertices ? n_vertices : 1));
    memcpy(buf, vertices, sizeof(real) * 3 * n_vertices); /* vertex_buffer = mesh.vertices.clone() */
    real *cur = vertices, *old = buf;
    for (uint32_t it = 0; it < iterations; ++it) {
        /* std::mem::swap(&mut vertex_buffer, &mut mesh.vertices): the new values are written over the older copy */
        real *tmp = cur;
        cur = old;
        old = tmp;
        for (uint64_t i = 0; i < n_vertices; ++i) {
            const real beta_eff = beta * weights[i];
            real sum[3] = {RC(0.0), RC(0.0), RC(0.0)};
            const uint64_t b = row_ptr[i], e = row_ptr[i + 1];
            for (uint64_t q = b; q < e; ++q)
                for (int d = 0; d < 3; ++d) sum[d] += old[3 * (uint64_t)nbrs[q] + d];
            if (e > b) {
                const real n = (real)(double)(e - b);
                for (int d = 0; d < 3; ++d) sum[d] /= n;
            }
            for (int d = 0; d < 3; ++d) cur[3 * i + d] = cur[3 * i + d] * (RC(1.0) - beta_eff) + sum[d] * beta_eff; /* :49 */
        }
    }
    if (cur != vertices) memcpy(vertices, cur, sizeof(real) * 3 * n_vertices);
    free(buf);
}

/* postprocessing.rs:55-96 (par_laplacian_smoothing_normals_inplace) */
void PFN(smooth_normals)(real *normals, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *nbrs, uint32_t iterations) {
    real *buf = (real *)calloc(3 * (n_vertices ? n_vertices : 1), sizeof(real));
    real *old = buf, *smoothed = normals;
    for (uint32_t it = 0; it < iterations; ++it) {
        real *tmp = old;
        old = smoothed;
        smoothed = tmp;
        for (uint64_t i = 0; i < n_vertices; ++i) {
            real s[3] = {RC(0.0), RC(0.0), RC(0.0)};
            for (uint64_t q = row_ptr[i]; q < row_ptr[i + 1]; ++q)
                for (int d = 0; d < 3; ++d) s[d] += old[3 * (uint64_t)nbrs[q] + d];
            const real norm = R_SQRT(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]); /* normalize_mut */
            for (int d = 0; d < 3; ++d) smoothed[3 * i + d] = s[d] / norm;
        }
    }
    if (smoothed != normals) memcpy(normals, smoothed, sizeof(real) * 3 * n_vertices);
    free(buf);
}

/* splashsurf/src/reconstruct.rs:1189-1204: distance-weighted neighbour count per particle, folded in list order */
void PFN(weighted_neighbor_counts)(const real *xyz, uint64_t n, const uint64_t *nb_ptr, const uint64_t *nb_idx, real h, real *out) {
    const real squared_r = h * h;
    for (uint64_t i = 0; i < n; ++i) {
        real acc = RC(0.0);
        for (uint64_t q = nb_ptr[i]; q < nb_ptr[i + 1]; ++q) {
            const real *pj = xyz + 3 * nb_idx[q], *pi = xyz + 3 * i;
            const real dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
            const real dist = dx * dx + dy * dy + dz * dz;
            real x = dist / squared_r;
            x = x < RC(0.0) ? RC(0.0) : (x > RC(1.0) ? RC(1.0) : x); /* clamp */
            acc = acc + (RC(1.0) - x);
        }
        out[i] = acc;
    }
}

/* splashsurf/src/reconstruct.rs:1219-1232: normalisation + smooth-step of the interpolated counts */
void PFN(smoothing_weights)(const real *wnn, uint64_t n, real normalization, real *out) {
    const real offset = RC(0.0);
    const real norm = normalization - offset;
    for (uint64_t i = 0; i < n; ++i) {
        real v = wnn[i] - offset;
        v = v > RC(0.0) ? v : RC(0.0); /* max */
        real x = v / norm;
        x = x < RC(1.0) ? x : RC(1.0); /* min */
        /* powi(5)*6 - powi(4)*15 + powi(3)*10; powi by repeated squaring (compiler-rt __powisf2/__powidf2) */
        const real x2 = x * x, x4 = x2 * x2;
        const real x5 = x * x4, x3 = x * x2;
        out[i] = x5 * RC(6.0) - x4 * RC(15.0) + x3 * RC(10.0);
    }
}

/* ---- SPH interpolation (sph_interpolation.rs) over a uniform cell grid (see the header for the order) ---- */
typedef struct {
    real origin[3];
    real h;
    int64_t nc[3];
    uint32_t *cell_start;
    uint32_t *items;
} sph_grid;

static void sph_grid_build(sph_grid *g, const real *xyz, uint64_t n, real h) {
    real mn[3] = {RC(0.0), RC(0.0), RC(0.0)}, mx[3] = {RC(0.0), RC(0.0), RC(0.0)};
    for (uint64_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            const real v = xyz[3 * i + d];
            if (i == 0 || v < mn[d]) mn[d] = v;
            if (i == 0 || v > mx[d]) mx[d] = v;
        }
    g->h = h;
    for (int d = 0; d < 3; ++d) {
        g->origin[d] = R_FLOOR(mn[d] / h) * h - h; /* one cell of padding */
        g->nc[d] = (int64_t)(double)R_FLOOR((mx[d] - g->origin[d]) / h) + 2;
    }
    const size_t ncell = (size_t)(g->nc[0] * g->nc[1] * g->nc[2]);
    g->cell_start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    g->items = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *cell_of = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint64_t i = 0; i < n; ++i) {
        int64_t c[3];
        for (int d = 0; d < 3; ++d) c[d] = (int64_t)(double)R_FLOOR((xyz[3 * i + d] - g->origin[d]) / h);
        cell_of[i] = (uint32_t)((c[0] * g->nc[1] + c[1]) * g->nc[2] + c[2]);
        g->cell_start[cell_of[i] + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) g->cell_start[c + 1] += g->cell_start[c];
    uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * (ncell ? ncell : 1));
    memcpy(cursor, g->cell_start, sizeof(uint32_t) * ncell);
    for (uint64_t i = 0; i < n; ++i) g->items[cursor[cell_of[i]]++] = (uint32_t)i;
    free(cursor);
    free(cell_of);
}

static void sph_grid_free(sph_grid *g) {
    free(g->cell_start);
    free(g->items);
}

static real cubic_function(real q) { /* kernel.rs:71-81 */
    const real pi = RC(3.14159265358979323846);
    if (q < RC(1.0)) return (RC(3.0) / (RC(2.0) * pi)) * ((RC(2.0) / RC(3.0)) - q * q + RC(0.5) * q * q * q);
    if (q < RC(2.0)) {
        const real x = RC(2.0) - q;
        return (RC(1.0) / (RC(4.0) * pi)) * x * x * x;
    }
    return RC(0.0);
}

static real cubic_function_dq(real q) { /* kernel.rs:84-94 */
    const real pi = RC(3.14159265358979323846);
    if (q < RC(1.0)) return (RC(3.0) / (RC(4.0) * pi)) * (RC(-4.0) * q + RC(3.0) * q * q);
    if (q < RC(2.0)) {
        const real x = RC(2.0) - q;
        return -(RC(3.0) / (RC(4.0) * pi)) * x * x;
    }
    return RC(0.0);
}

/* interpolate_quantity_inplace (sph_interpolation.rs:205-259) for `dim` components per particle (1: scalar, 3: vector) */
void PFN(sph_interpolate)(const real *xyz, const real *rho, uint64_t n, real rest_mass, real h, const real *values, int dim, const real *points,
                          uint64_t n_points, int first_order_correction, real *out) {
    sph_grid g;
    sph_grid_build(&g, xyz, n, h);
    const real squared_support = h * h;
    const real sigma = RC(8.0) / (h * h * h);
    const real enable = first_order_correction ? RC(1.0) : RC(0.0);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)n_points; ++p) {
        const real *x = points + 3 * p;
        real acc[4] = {RC(0.0), RC(0.0), RC(0.0), RC(0.0)};
        real correction = RC(0.0);
        int64_t c[3];
        for (int d = 0; d < 3; ++d) c[d] = (int64_t)(double)R_FLOOR((x[d] - g.origin[d]) / h);
        for (int64_t cx = c[0] - 1; cx <= c[0] + 1; ++cx)
            for (int64_t cy = c[1] - 1; cy <= c[1] + 1; ++cy)
                for (int64_t cz = c[2] - 1; cz <= c[2] + 1; ++cz) {
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= g.nc[0] || cy >= g.nc[1] || cz >= g.nc[2]) continue;
                    const size_t f = (size_t)((cx * g.nc[1] + cy) * g.nc[2] + cz);
                    for (uint32_t q = g.cell_start[f]; q < g.cell_start[f + 1]; ++q) {
                        const uint32_t j = g.items[q];
                        const real dx = xyz[3 * (size_t)j] - x[0], dy = xyz[3 * (size_t)j + 1] - x[1], dz = xyz[3 * (size_t)j + 2] - x[2];
                        const real d2 = dx * dx + dy * dy + dz * dz;
                        if (!(d2 <= squared_support)) continue; /* locate_within_distance: distance_2 <= max_squared_radius */
                        const real vol = rest_mass / rho[j]; /* :299 */
                        const real r = R_SQRT(d2);
                        const real w = sigma * cubic_function((r + r) / h);
                        const real vw = vol * w;
                        for (int k = 0; k < dim; ++k) acc[k] += values[(size_t)dim * j + k] * vw; /* A_j.scale(vol_j * W_ij) */
                        correction += vw;
                    }
                }
        const real factor = enable * (RC(1.0) / correction) + (RC(1.0) - enable); /* :253-255 */
        for (int k = 0; k < dim; ++k) out[(size_t)dim * p + k] = acc[k] * factor;
    }
    sph_grid_free(&g);
}

/* interpolate_normals_inplace (sph_interpolation.rs:72-113) */
void PFN(sph_normals)(const real *xyz, const real *rho, uint64_t n, real rest_mass, real h, const real *points, uint64_t n_points, real *out) {
    sph_grid g;
    sph_grid_build(&g, xyz, n, h);
    const real squared_support = h * h;
    const real sigma = RC(8.0) / (h * h * h);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)n_points; ++p) {
        const real *x = points + 3 * p;
        real grad[3] = {RC(0.0), RC(0.0), RC(0.0)};
        int64_t c[3];
        for (int d = 0; d < 3; ++d) c[d] = (int64_t)(double)R_FLOOR((x[d] - g.origin[d]) / h);
        for (int64_t cx = c[0] - 1; cx <= c[0] + 1; ++cx)
            for (int64_t cy = c[1] - 1; cy <= c[1] + 1; ++cy)
                for (int64_t cz = c[2] - 1; cz <= c[2] + 1; ++cz) {
                    if (cx < 0 || cy < 0 || cz < 0 || cx >= g.nc[0] || cy >= g.nc[1] || cz >= g.nc[2]) continue;
                    const size_t f = (size_t)((cx * g.nc[1] + cy) * g.nc[2] + cz);
                    for (uint32_t q = g.cell_start[f]; q < g.cell_start[f + 1]; ++q) {
                        const uint32_t j = g.items[q];
                        const real dx[3] = {xyz[3 * (size_t)j] - x[0], xyz[3 * (size_t)j + 1] - x[1], xyz[3 * (size_t)j + 2] - x[2]};
                        const real d2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
                        if (!(d2 <= squared_support)) continue;
                        const real vol = rest_mass / rho[j];
                        const real r = R_SQRT(d2); /* dx.norm() */
                        const real q_ = (r + r) / h;
                        const real gnorm = sigma * cubic_function_dq(q_) * ((RC(1.0) + RC(1.0)) / h); /* kernel.rs:132-139 */
                        for (int d = 0; d < 3; ++d) grad[d] += ((dx[d] / r) * gnorm) * vol; /* :103-104 */
                    }
                }
        const real norm = R_SQRT(grad[0] * grad[0] + grad[1] * grad[1] + grad[2] * grad[2]); /* Unit::new_normalize */
        for (int d = 0; d < 3; ++d) out[3 * p + d] = grad[d] / norm;
    }
    sph_grid_free(&g);
}
