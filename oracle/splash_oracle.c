/* splash_oracle.c -- CPU ORACLE (test infrastructure, NOT the product; see splash_oracle.h).
 *
 * Restates, in plain C99 with IEEE f32 arithmetic (compile with -ffp-contract=off, no fast-math),
 * the reference's subdomain-grid reconstruction path.  All `file:line` citations are relative to
 * /root/reference/splashsurf_lib/src/.  Index type I = i64, Real = f32 (the instantiation used by
 * the reference's CLI and Python binding, splashsurf/src/reconstruct.rs:982-1007).
 *
 * Deliberate, documented deviation: the reference iterates subdomains in dashmap order (machine
 * dependent, dense_subdomains.rs:388-425); this oracle iterates them in ascending flat subdomain
 * index.  This only changes the ORDER of output vertices/triangles and which of two ulp-different
 * coordinate sets wins for a vertex on a subdomain face ("first patch wins",
 * dense_subdomains.rs:1707-1716); the reference's own output has the same freedom.
 */
#include "splash_oracle.h"

/* One source, two translation units: -DSO_F64 builds the f64 instantiation (so64_*). */
#include <float.h>
#ifdef SO_F64
typedef double real;
#define SOT(name) so64_##name
#define SOFN(name) so64_##name
#define R_FLOOR floor
#define R_CEIL ceil
#define R_SQRT sqrt
#define R_EPSILON DBL_EPSILON
#else
typedef float real;
#define SOT(name) so_##name
#define SOFN(name) so_##name
#define R_FLOOR floorf
#define R_CEIL ceilf
#define R_SQRT sqrtf
#define R_EPSILON FLT_EPSILON
#endif
/* literal of the Real type: the reference writes f64 literals converted with R::from_float */
#define RC(x) ((real)(x))

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------
 * MC table (marching_cubes/marching_cubes_lut.rs:44-344), in emitted (winding-flipped) order.
 * ------------------------------------------------------------------------------------------ */
static const int8_t MC_TABLE[256][16] = {
#include "../splashsurf_amd/csrc/mc_table.inc"
};
const int8_t *SOFN(mc_table)(void) { return &MC_TABLE[0][0]; }

/* uniform_grid.rs:825-834 */
static const int8_t CELL_LOCAL_POINT_COORDS[8][3] = {
    {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
/* uniform_grid.rs:856-869: local edge -> (origin local point, axis) */
static const int8_t CELL_LOCAL_EDGES[12][2] = {{0, 0}, {1, 1}, {3, 0}, {0, 1}, {4, 0}, {5, 1},
                                               {7, 0}, {4, 1}, {0, 2}, {1, 2}, {2, 2}, {3, 2}};

/* ------------------------------------------------------------------------------------------
 * Cubic spline kernel, scalar path (kernel.rs:58-107)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    real h;
    real sigma;
} cubic_kernel;

static cubic_kernel kernel_new(real h) { /* kernel.rs:60-68 */
    cubic_kernel k;
    k.h = h;
    k.sigma = RC(8.0) / (h * h * h);
    return k;
}

static real cubic_function(real q) { /* kernel.rs:71-81 */
    const real pi = RC(3.14159265358979323846);
    if (q < RC(1.0)) {
        return (RC(3.0) / (RC(2.0) * pi)) * ((RC(2.0) / RC(3.0)) - q * q + RC(0.5) * q * q * q);
    } else if (q < RC(2.0)) {
        real x = RC(2.0) - q;
        return (RC(1.0) / (RC(4.0) * pi)) * x * x * x;
    } else {
        return RC(0.0);
    }
}

static real kernel_evaluate(const cubic_kernel *k, real r) { /* kernel.rs:103-106 */
    real q = (r + r) / k->h;
    return k->sigma * cubic_function(q);
}

real SOFN(cubic_kernel_evaluate)(real h, real r) {
    cubic_kernel k = kernel_new(h);
    return kernel_evaluate(&k, r);
}

/* ------------------------------------------------------------------------------------------
 * Uniform grid (uniform_grid.rs)
 * ------------------------------------------------------------------------------------------ */
static void grid_new(SOT(grid) *g, const real min[3], const int64_t n_cells[3], real cs) {
    /* uniform_grid.rs:203-232, 662-674 */
    for (int d = 0; d < 3; ++d) {
        g->aabb_min[d] = min[d];
        g->n_cells[d] = n_cells[d];
        g->n_points[d] = n_cells[d] + 1;
        g->aabb_max[d] = min[d] + cs * (real)(double)n_cells[d];
    }
    g->cell_size = cs;
}

/* returns 0 ok, 1 invalid cell size, 2 degenerate, 3 inconsistent (uniform_grid.rs:175-201) */
static int grid_from_aabb(SOT(grid) *g, const real amin[3], const real amax[3], real cs) {
    if (!(cs > RC(0.0))) return 1;
    if (amin[0] == amax[0] && amin[1] == amax[1] && amin[2] == amax[2]) return 2; /* aabb.rs:159-161 */
    if (!(amin[0] <= amax[0] && amin[1] <= amax[1] && amin[2] <= amax[2])) return 3; /* aabb.rs:147-149 */
    real aligned_min[3];
    int64_t n_cells[3];
    for (int d = 0; d < 3; ++d) {
        aligned_min[d] = R_FLOOR(amin[d] / cs) * cs;
        real n_cells_real = (amax[d] - aligned_min[d]) / cs;
        int64_t n = (int64_t)(double)R_CEIL(n_cells_real); /* uniform_grid.rs:647-655 */
        n_cells[d] = n < 1 ? 1 : n;
    }
    grid_new(g, aligned_min, n_cells, cs);
    return 0;
}

static inline real grid_point_coord(const SOT(grid) *g, int64_t i, int d) { /* uniform_grid.rs:418-425 */
    return g->aabb_min[d] + (real)(double)i * g->cell_size;
}

static inline void grid_enclosing_cell(const SOT(grid) *g, const real p[3], int64_t ijk[3]) {
    /* uniform_grid.rs:444-451 */
    for (int d = 0; d < 3; ++d) {
        real normalized = (p[d] - g->aabb_min[d]) / g->cell_size;
        ijk[d] = (int64_t)(double)R_FLOOR(normalized);
    }
}

static inline int grid_cell_exists(const SOT(grid) *g, const int64_t ijk[3]) { /* uniform_grid.rs:311-319 */
    return ijk[0] >= 0 && ijk[1] >= 0 && ijk[2] >= 0 && ijk[0] < g->n_cells[0] &&
           ijk[1] < g->n_cells[1] && ijk[2] < g->n_cells[2];
}

static inline int64_t grid_flatten_cell(const SOT(grid) *g, const int64_t ijk[3]) { /* uniform_grid.rs:361-365 */
    return ijk[0] * g->n_cells[1] * g->n_cells[2] + ijk[1] * g->n_cells[2] + ijk[2];
}

static inline void grid_unflatten_cell(const SOT(grid) *g, int64_t flat, int64_t ijk[3]) { /* uniform_grid.rs:399-407 */
    int64_t nyz = g->n_cells[1] * g->n_cells[2];
    ijk[0] = flat / nyz;
    ijk[1] = (flat - ijk[0] * nyz) / g->n_cells[2];
    ijk[2] = flat - ijk[0] * nyz - ijk[1] * g->n_cells[2];
}

/* ------------------------------------------------------------------------------------------
 * Grid set-up (lib.rs:476-516, density_map.rs:551-580)
 * ------------------------------------------------------------------------------------------ */
static int grid_for_particle_aabb(const real pmin[3], const real pmax[3], const SOT(params) *P, SOT(grid) *out) {
    /* lib.rs:496-515 for a known particle AABB */
    real amin[3], amax[3];
    for (int d = 0; d < 3; ++d) {
        amin[d] = pmin[d] - P->particle_radius;
        amax[d] = pmax[d] + P->particle_radius;
    }
    real half_supported_cells_real = R_CEIL(P->compact_support_radius / P->cube_size);
    const real eps_sqrt = R_SQRT(R_EPSILON);
    real kernel_margin = P->cube_size * half_supported_cells_real * (RC(1.0) + eps_sqrt);
    for (int d = 0; d < 3; ++d) {
        amin[d] -= kernel_margin;
        amax[d] += kernel_margin;
    }
    return grid_from_aabb(out, amin, amax, P->cube_size);
}

static int grid_for_reconstruction(const real *xyz, uint64_t n, const SOT(params) *P, SOT(grid) *out) {
    real amin[3], amax[3];
    if (P->has_particle_aabb) { /* lib.rs:484-485 */
        for (int d = 0; d < 3; ++d) {
            amin[d] = P->aabb_min[d];
            amax[d] = P->aabb_max[d];
        }
    } else {
        /* aabb.rs:28-52: empty -> zeros */
        if (n == 0) {
            for (int d = 0; d < 3; ++d) amin[d] = amax[d] = RC(0.0);
        } else {
            for (int d = 0; d < 3; ++d) amin[d] = amax[d] = xyz[d];
            for (uint64_t i = 1; i < n; ++i)
                for (int d = 0; d < 3; ++d) {
                    real v = xyz[3 * i + d];
                    if (v < amin[d]) amin[d] = v;
                    if (v > amax[d]) amax[d] = v;
                }
        }
        for (int d = 0; d < 3; ++d) { /* lib.rs:496 */
            amin[d] -= P->particle_radius;
            amax[d] += P->particle_radius;
        }
    }
    /* density_map.rs:551-580 */
    real half_supported_cells_real = R_CEIL(P->compact_support_radius / P->cube_size);
    const real eps_sqrt = R_SQRT(R_EPSILON);
    real kernel_margin = P->cube_size * half_supported_cells_real * (RC(1.0) + eps_sqrt);
    for (int d = 0; d < 3; ++d) { /* lib.rs:513 */
        amin[d] -= kernel_margin;
        amax[d] += kernel_margin;
    }
    return grid_from_aabb(out, amin, amax, P->cube_size);
}

/* ------------------------------------------------------------------------------------------
 * Parameters of the subdomain grid (dense_subdomains.rs:89-244)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    real particle_rest_mass;
    real h;
    real threshold;
    real cube_size;
    int64_t subdomain_cubes;
    real ghost_margin;
    SOT(grid) global_mc_grid;
    SOT(grid) subdomain_grid;
} sd_params;

static int initialize_parameters(const SOT(params) *P, const SOT(grid) *initial_grid, sd_params *S) {
    int64_t nc = P->subdomain_num_cubes_per_dim;
    real d = P->particle_radius + P->particle_radius; /* kernel.rs:28-30 */
    real rest_volume = d * d * d;
    S->particle_rest_mass = rest_volume * P->rest_density; /* dense_subdomains.rs:117-118 */
    S->h = P->compact_support_radius;
    S->threshold = P->iso_surface_threshold;
    S->cube_size = P->cube_size;
    S->subdomain_cubes = nc;
    S->ghost_margin = R_CEIL(P->compact_support_radius / P->cube_size) * P->cube_size * RC(1.01); /* :120-121 */
    int64_t num_sub[3], num_cells[3];
    for (int k = 0; k < 3; ++k) { /* :168-181, 2129-2131 */
        int64_t c = initial_grid->n_cells[k];
        num_sub[k] = c / nc + ((c % nc) < 1 ? (c % nc) : 1);
        num_cells[k] = num_sub[k] * nc;
    }
    grid_new(&S->global_mc_grid, initial_grid->aabb_min, num_cells, P->cube_size); /* :183-188 */
    real subdomain_size = P->cube_size * (real)(double)nc;                      /* :207 */
    grid_new(&S->subdomain_grid, S->global_mc_grid.aabb_min, num_sub, subdomain_size); /* :209-213 */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Ghost-margin classification (dense_subdomains.rs:1810-1905)
 * ------------------------------------------------------------------------------------------ */
static int classify_particle(const SOT(grid) *sg, real margin, const real p[3], int64_t *out, int cap) {
    int64_t sub[3];
    grid_enclosing_cell(sg, p, sub);
    if (!grid_cell_exists(sg, sub)) return 0; /* :1819-1822 */
    real dx = sg->cell_size;
    int r = (int)(int64_t)(double)R_CEIL(margin / dx); /* :1827-1832 */
    real min_corner[3], max_corner[3];                /* uniform_grid.rs:454-467 */
    for (int d = 0; d < 3; ++d) {
        min_corner[d] = grid_point_coord(sg, sub[d], d);
        max_corner[d] = grid_point_coord(sg, sub[d] + 1, d);
    }
    int count = 0;
    for (int i = -r; i <= r; ++i)
        for (int j = -r; j <= r; ++j)
            for (int k = -r; k <= r; ++k) {
                int steps[3] = {i, j, k};
                int in_margin = 1;
                for (int d = 0; d < 3 && in_margin; ++d) { /* :1844-1856 */
                    int step = steps[d];
                    real off = (real)((step < 0 ? -step : step) - 1);
                    if (step > 0)
                        in_margin = ((max_corner[d] + off * dx) - p[d]) < margin;
                    else if (step < 0)
                        in_margin = (p[d] - (min_corner[d] - off * dx)) < margin;
                }
                if (!in_margin) continue;
                int64_t nb[3] = {sub[0] + i, sub[1] + j, sub[2] + k};
                if (!grid_cell_exists(sg, nb)) continue; /* :1895-1900 */
                if (count < cap) out[count] = grid_flatten_cell(sg, nb);
                ++count;
            }
    return count;
}

int SOFN(classify_particle)(const SOT(grid) *sg, real margin, const real p[3], int64_t *out, int cap) {
    return classify_particle(sg, margin, p, out, cap);
}

/* ------------------------------------------------------------------------------------------
 * Decomposition (dense_subdomains.rs:349-494): per-subdomain particle index lists, ascending.
 * Subdomains are stored in ascending flat index (documented deviation, see file header).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n_sub;         /* occupied subdomains */
    int64_t *flat_index;   /* [n_sub] */
    uint64_t *offsets;     /* [n_sub+1] into particles */
    uint32_t *particles;   /* concatenated ascending particle indices */
} subdomains_t;

static void subdomains_free(subdomains_t *s) {
    free(s->flat_index);
    free(s->offsets);
    free(s->particles);
    memset(s, 0, sizeof(*s));
}

/* optional restriction to a box of subdomains (multi-process shard); NULL = all */
static int classify_particle_boxed(const SOT(grid) *sg, real margin, const real p[3], int64_t *out, int cap, const int64_t *lo,
                                   const int64_t *hi) {
    int m = classify_particle(sg, margin, p, out, cap);
    if (!lo) return m;
    if (m > cap) m = cap;
    int k = 0;
    for (int q = 0; q < m; ++q) {
        int64_t ijk[3];
        grid_unflatten_cell(sg, out[q], ijk);
        if (ijk[0] >= lo[0] && ijk[1] >= lo[1] && ijk[2] >= lo[2] && ijk[0] < hi[0] && ijk[1] < hi[1] && ijk[2] < hi[2]) out[k++] = out[q];
    }
    return k;
}

static int decomposition_boxed(const sd_params *S, const real *xyz, uint64_t n, subdomains_t *out, int nthreads, const int64_t *box_lo,
                               const int64_t *box_hi);

static int decomposition(const sd_params *S, const real *xyz, uint64_t n, subdomains_t *out, int nthreads) {
    return decomposition_boxed(S, xyz, n, out, nthreads, NULL, NULL);
}

static int decomposition_boxed(const sd_params *S, const real *xyz, uint64_t n, subdomains_t *out, int nthreads, const int64_t *box_lo,
                               const int64_t *box_hi) {
    const SOT(grid) *sg = &S->subdomain_grid;
    int64_t total = sg->n_cells[0] * sg->n_cells[1] * sg->n_cells[2];
    int cap = 27;
    {
        int r = (int)(int64_t)(double)R_CEIL(S->ghost_margin / sg->cell_size);
        cap = (2 * r + 1) * (2 * r + 1) * (2 * r + 1);
    }
    if (nthreads < 1) nthreads = 1;
    /* counts[t][s] */
    uint64_t *counts = (uint64_t *)calloc((size_t)nthreads * (size_t)total, sizeof(uint64_t));
    if (!counts) return 4;
    uint64_t chunk = (n + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        int64_t *buf = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        uint64_t lo = (uint64_t)t * chunk, hi = lo + chunk;
        if (hi > n) hi = n;
        uint64_t *c = counts + (size_t)t * (size_t)total;
        for (uint64_t i = lo; i < hi; ++i) {
            int m = classify_particle_boxed(sg, S->ghost_margin, xyz + 3 * i, buf, cap, box_lo, box_hi);
            for (int q = 0; q < m; ++q) c[buf[q]]++;
        }
        free(buf);
    }
    /* occupied subdomains in ascending flat order; offsets per (subdomain, thread) */
    int64_t n_sub = 0;
    for (int64_t s = 0; s < total; ++s) {
        uint64_t tot = 0;
        for (int t = 0; t < nthreads; ++t) tot += counts[(size_t)t * (size_t)total + (size_t)s];
        if (tot) ++n_sub;
    }
    out->n_sub = n_sub;
    out->flat_index = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_sub > 0 ? n_sub : 1));
    out->offsets = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_sub + 1));
    int64_t *compressed = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total > 0 ? total : 1));
    uint64_t run = 0;
    int64_t k = 0;
    for (int64_t s = 0; s < total; ++s) {
        uint64_t tot = 0;
        for (int t = 0; t < nthreads; ++t) {
            uint64_t c = counts[(size_t)t * (size_t)total + (size_t)s];
            counts[(size_t)t * (size_t)total + (size_t)s] = run + tot; /* becomes write cursor */
            tot += c;
        }
        if (tot) {
            compressed[s] = k;
            out->flat_index[k] = s;
            out->offsets[k] = run;
            ++k;
            run += tot;
        } else {
            compressed[s] = -1;
        }
    }
    out->offsets[n_sub] = run;
    out->particles = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(run > 0 ? run : 1));
#pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        int64_t *buf = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
        uint64_t lo = (uint64_t)t * chunk, hi = lo + chunk;
        if (hi > n) hi = n;
        uint64_t *c = counts + (size_t)t * (size_t)total;
        for (uint64_t i = lo; i < hi; ++i) {
            int m = classify_particle_boxed(sg, S->ghost_margin, xyz + 3 * i, buf, cap, box_lo, box_hi);
            for (int q = 0; q < m; ++q) out->particles[c[buf[q]]++] = (uint32_t)i;
        }
        free(buf);
    }
    free(compressed);
    free(counts);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Per-subdomain neighbourhood search + density (dense_subdomains.rs:496-646,
 * neighborhood_search.rs:345-438, density_map.rs:150-186)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    real *pos;        /* gathered positions [P*3] */
    real *rho;        /* gathered densities [P] */
    uint32_t *cell_of; /* [P] flat search cell */
    uint32_t *cell_start; /* [ncell+1] */
    uint32_t *cell_items; /* [P] local indices grouped by cell, ascending in each */
    size_t cap_p, cap_cells;
    real *levelset;      /* (n+1)^3 */
    int32_t *edge_to_vertex; /* 3*(n+1)^3 */
    uint32_t *touched;       /* list of edge slots set, for cheap reset */
    size_t n_touched, cap_touched;
} workspace_t;

static void ws_reserve_particles(workspace_t *w, size_t p) {
    if (p > w->cap_p) {
        size_t c = p + p / 2 + 64;
        w->pos = (real *)realloc(w->pos, sizeof(real) * 3 * c);
        w->rho = (real *)realloc(w->rho, sizeof(real) * c);
        w->cell_of = (uint32_t *)realloc(w->cell_of, sizeof(uint32_t) * c);
        w->cell_items = (uint32_t *)realloc(w->cell_items, sizeof(uint32_t) * c);
        w->cap_p = c;
    }
}

static void ws_free(workspace_t *w) {
    free(w->pos);
    free(w->rho);
    free(w->cell_of);
    free(w->cell_start);
    free(w->cell_items);
    free(w->levelset);
    free(w->edge_to_vertex);
    free(w->touched);
    memset(w, 0, sizeof(*w));
}

static void subdomain_aabb(const sd_params *S, const int64_t sub[3], real amin[3], real amax[3]) {
    /* uniform_grid.rs:454-467 */
    for (int d = 0; d < 3; ++d) {
        amin[d] = grid_point_coord(&S->subdomain_grid, sub[d], d);
        amax[d] = grid_point_coord(&S->subdomain_grid, sub[d] + 1, d);
    }
}

static void subdomain_density(const sd_params *S, const real *xyz, const uint32_t *idx, size_t P,
                              int64_t flat_sub, workspace_t *w, real *global_rho, uint32_t **nb_lists, uint32_t *nb_counts) {
    ws_reserve_particles(w, P);
    for (size_t a = 0; a < P; ++a) { /* gather_subdomain_data :545 */
        const real *p = xyz + 3 * (size_t)idx[a];
        w->pos[3 * a] = p[0];
        w->pos[3 * a + 1] = p[1];
        w->pos[3 * a + 2] = p[2];
    }
    int64_t sub[3];
    grid_unflatten_cell(&S->subdomain_grid, flat_sub, sub);
    real amin[3], amax[3], mmin[3], mmax[3];
    subdomain_aabb(S, sub, amin, amax);
    real grow = S->ghost_margin * RC(1.5); /* :560-565 */
    for (int d = 0; d < 3; ++d) {
        mmin[d] = amin[d] - grow;
        mmax[d] = amax[d] + grow;
    }
    SOT(grid) sgrid; /* neighborhood_search.rs:370 */
    int rc = grid_from_aabb(&sgrid, mmin, mmax, S->h);
    if (rc != 0) {
        fprintf(stderr, "oracle: failed to construct search grid\n");
        abort();
    }
    size_t ncell = (size_t)(sgrid.n_cells[0] * sgrid.n_cells[1] * sgrid.n_cells[2]);
    if (ncell + 1 > w->cap_cells) {
        w->cell_start = (uint32_t *)realloc(w->cell_start, sizeof(uint32_t) * (ncell + 1));
        w->cap_cells = ncell + 1;
    }
    memset(w->cell_start, 0, sizeof(uint32_t) * (ncell + 1));
    /* cell -> particles map (neighborhood_search.rs:679-710); the hash map is replaced by a dense
       counting sort, which yields the same per-cell ascending insertion order. */
    for (size_t a = 0; a < P; ++a) {
        int64_t c[3];
        grid_enclosing_cell(&sgrid, w->pos + 3 * a, c);
        if (!grid_cell_exists(&sgrid, c)) {
            fprintf(stderr, "oracle: particle outside search grid (reference would panic)\n");
            abort();
        }
        uint32_t f = (uint32_t)grid_flatten_cell(&sgrid, c);
        w->cell_of[a] = f;
        w->cell_start[f + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) w->cell_start[c + 1] += w->cell_start[c];
    {
        /* stable fill */
        uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * (ncell ? ncell : 1));
        memcpy(cursor, w->cell_start, sizeof(uint32_t) * ncell);
        for (size_t a = 0; a < P; ++a) w->cell_items[cursor[w->cell_of[a]]++] = (uint32_t)a;
        free(cursor);
    }
    cubic_kernel K = kernel_new(S->h);
    real h2 = S->h * S->h; /* neighborhood_search.rs:367 */
    real w0 = kernel_evaluate(&K, RC(0.0));
    for (size_t a = 0; a < P; ++a) {
        const real *pi = w->pos + 3 * a;
        /* is_inside: half-open AABB test (:567-576, aabb.rs:220-222) */
        int inside = pi[0] >= amin[0] && pi[1] >= amin[1] && pi[2] >= amin[2] && pi[0] < amax[0] &&
                     pi[1] < amax[1] && pi[2] < amax[2];
        if (!inside) continue;
        int64_t ci[3];
        grid_enclosing_cell(&sgrid, pi, ci);
        real density = w0; /* density_map.rs:173 */
        uint32_t *nbl = NULL;
        size_t nbn = 0, nbcap = 0;
        /* 26 adjacent cells in iproduct order (uniform_grid.rs:614-643), then the own cell
           (neighborhood_search.rs:400-405) */
        for (int pass = 0; pass < 2; ++pass) {
            for (int sx = -1; sx <= 1; ++sx)
                for (int sy = -1; sy <= 1; ++sy)
                    for (int sz = -1; sz <= 1; ++sz) {
                        int is_center = (sx == 0 && sy == 0 && sz == 0);
                        if ((pass == 0) == is_center) continue;
                        int64_t c[3] = {ci[0] + sx, ci[1] + sy, ci[2] + sz};
                        if (!grid_cell_exists(&sgrid, c)) continue;
                        size_t f = (size_t)grid_flatten_cell(&sgrid, c);
                        for (uint32_t q = w->cell_start[f]; q < w->cell_start[f + 1]; ++q) {
                            uint32_t b = w->cell_items[q];
                            const real *pj = w->pos + 3 * (size_t)b;
                            real dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];
                            real d2 = dx * dx + dy * dy + dz * dz; /* nalgebra norm_squared: (x2+y2)+z2 */
                            if (b != (uint32_t)a && d2 < h2) { /* neighborhood_search.rs:431 */
                                real r = R_SQRT(d2);          /* density_map.rs:179 */
                                density += kernel_evaluate(&K, r);
                                if (nb_lists) { /* dense_subdomains.rs:617-639: local -> global index */
                                    if (nbn == nbcap) {
                                        nbcap = nbcap ? nbcap * 2 : 64;
                                        nbl = (uint32_t *)realloc(nbl, sizeof(uint32_t) * nbcap);
                                    }
                                    nbl[nbn++] = idx[b];
                                }
                            }
                        }
                    }
        }
        density *= S->particle_rest_mass; /* density_map.rs:182 */
        global_rho[idx[a]] = density;     /* :596-614 */
        if (nb_lists) {
            nb_lists[idx[a]] = nbl;
            nb_counts[idx[a]] = (uint32_t)nbn;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Level-set evaluation, scalar loop (dense_subdomains.rs:660-693, 784-847)
 * ------------------------------------------------------------------------------------------ */
static void density_grid_loop_scalar(const sd_params *S, const int64_t sub[3], const real *pos,
                                     const real *rho, size_t P, real *levelset) {
    const int64_t n = S->subdomain_cubes, np = n + 1;
    real amin[3], amax[3];
    subdomain_aabb(S, sub, amin, amax);
    SOT(grid) mc; /* :1383-1388 */
    int64_t nc3[3] = {n, n, n};
    grid_new(&mc, amin, nc3, S->cube_size);
    const int64_t cube_radius = (int64_t)(double)R_CEIL(S->h / S->cube_size); /* :1228 */
    const real support_sq_margin = (S->h * S->h) * RC(1.01);                  /* :1224-1226 */
    cubic_kernel K = kernel_new(S->h);
    const SOT(grid) *gg = &S->global_mc_grid;
    for (size_t a = 0; a < P; ++a) {
        const real *p = pos + 3 * a;
        real rho_i = rho[a];
        int64_t cell[3], lo[3], hi[3];
        grid_enclosing_cell(&mc, p, cell); /* :671 */
        for (int d = 0; d < 3; ++d) {      /* :676-690 */
            int64_t l = cell[d] - cube_radius;
            if (l < 0) l = 0;
            if (l > np) l = np;
            int64_t u = cell[d] + cube_radius + 2;
            if (u > np) u = np;
            if (u < 0) u = 0;
            lo[d] = l;
            hi[d] = u;
        }
        for (int64_t i = lo[0]; i < hi[0]; ++i) {
            real gx = grid_point_coord(gg, sub[0] * n + i, 0); /* :817-826 */
            real dx = p[0] - gx;
            for (int64_t j = lo[1]; j < hi[1]; ++j) {
                real gy = grid_point_coord(gg, sub[1] * n + j, 1);
                real dy = p[1] - gy;
                for (int64_t k = lo[2]; k < hi[2]; ++k) {
                    real gz = grid_point_coord(gg, sub[2] * n + k, 2);
                    real dz = p[2] - gz;
                    real d2 = dx * dx + dy * dy + dz * dz; /* :828-829 */
                    if (d2 < support_sq_margin) {           /* :831 */
                        real v_i = S->particle_rest_mass / rho_i;
                        real r = R_SQRT(d2);
                        real w_ij = kernel_evaluate(&K, r);
                        levelset[(i * np + j) * np + k] += v_i * w_ij; /* :837-841 */
                    }
                }
            }
        }
    }
}

#ifndef SO_F64
/* ------------------------------------------------------------------------------------------
 * Level-set evaluation, AVX2+FMA loop (dense_subdomains.rs:991-1133) with CubicSplineKernelAvxF32
 * (kernel.rs:319-378), restated lane by lane with fmaf.  `uniform` != 0: mode 2 of so_params.enable_simd
 * (every lane takes the fused vector path).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    float inv_h, sigma, sigma2, sigma6, sigma12;
} avx_kernel;

static avx_kernel avx_kernel_new(float h) { /* kernel.rs:327-337 */
    avx_kernel k;
    const float pi = 3.14159265358979323846f; /* std::f32::consts::PI */
    k.inv_h = 1.0f / h;
    const float rrr = h * h * h;
    k.sigma = 8.0f / (pi * rrr);
    k.sigma2 = 2.0f * k.sigma;
    k.sigma6 = 6.0f * k.sigma;
    k.sigma12 = 12.0f * k.sigma;
    return k;
}

static inline float avx_kernel_evaluate(const avx_kernel *K, float r) { /* kernel.rs:341-377 */
    const float q = r * K->inv_h;
    float v = 1.0f - q;
    v = v > 0.0f ? v : 0.0f; /* _mm256_max_ps(v, zero) */
    const float v2 = v * v;
    const float v3 = v2 * v;
    const float res_outer = v3 * K->sigma2;
    float res_inner = K->sigma;
    res_inner = fmaf(-v, K->sigma6, res_inner);  /* _mm256_fnmadd_ps */
    res_inner = fmaf(v2, K->sigma12, res_inner); /* _mm256_fmadd_ps */
    res_inner = fmaf(-v3, K->sigma6, res_inner);
    return (q <= 0.5f) ? res_inner : res_outer; /* _CMP_LE_OQ + blendv */
}

float so_avx_kernel_evaluate(float h, float r) {
    const avx_kernel K = avx_kernel_new(h);
    return avx_kernel_evaluate(&K, r);
}

static void density_grid_loop_avx(const sd_params *S, const int64_t sub[3], const float *pos, const float *rho, size_t P,
                                  float *levelset, int uniform) {
    const int64_t n = S->subdomain_cubes, np = n + 1;
    float amin[3], amax[3];
    subdomain_aabb(S, sub, amin, amax);
    SOT(grid) mc;
    int64_t nc3[3] = {n, n, n};
    grid_new(&mc, amin, nc3, S->cube_size);
    const int64_t cube_radius = (int64_t)(double)R_CEIL(S->h / S->cube_size);
    const avx_kernel K = avx_kernel_new(S->h); /* :1023 */
    const SOT(grid) *gg = &S->global_mc_grid;
    const float cube_size = gg->cell_size;     /* :1035 */
    const float support_sq = S->h * S->h;      /* :1037-1038 */
    for (size_t a = 0; a < P; ++a) {
        const float *p = pos + 3 * a;
        const float v_i = S->particle_rest_mass / rho[a]; /* :1045 */
        int64_t cell[3], lo[3], hi[3];
        grid_enclosing_cell(&mc, p, cell);
        for (int d = 0; d < 3; ++d) { /* particle_influence_aabb, :660-693 */
            int64_t l = cell[d] - cube_radius;
            if (l < 0) l = 0;
            if (l > np) l = np;
            int64_t u = cell[d] + cube_radius + 2;
            if (u > np) u = np;
            if (u < 0) u = 0;
            lo[d] = l;
            hi[d] = u;
        }
        if (hi[2] < lo[2]) continue;
        const int64_t remainder = uniform ? 0 : (hi[2] - lo[2]) % 8; /* :1051-1052 */
        const int64_t upper_k_aligned = hi[2] - remainder;
        for (int64_t i = lo[0]; i < hi[0]; ++i) {
            const int32_t global_i = (int32_t)sub[0] * (int32_t)n + (int32_t)i; /* :1110 */
            const float grid_x = (float)global_i * cube_size + gg->aabb_min[0]; /* :1113, scalar Rust: mul then add */
            const float dx = p[0] - grid_x;
            for (int64_t j = lo[1]; j < hi[1]; ++j) {
                const int32_t global_j = (int32_t)sub[1] * (int32_t)n + (int32_t)j;
                const float grid_y = (float)global_j * cube_size + gg->aabb_min[1];
                const float dy = p[1] - grid_y;
                for (int64_t k = lo[2]; k < hi[2]; ++k) {
                    const int32_t global_k = (int32_t)sub[2] * (int32_t)n + (int32_t)k;     /* :1067-1068 */
                    const float grid_z = fmaf((float)global_k, cube_size, gg->aabb_min[2]); /* :1069 _mm256_fmadd_ps */
                    const float dz = p[2] - grid_z;
                    const float dist_sq = fmaf(dz, dz, fmaf(dx, dx, dy * dy));               /* :1077-1080 */
                    if (!(dist_sq < support_sq)) continue; /* :1083-1086, 1091: masked lanes contribute +0 */
                    const float w = avx_kernel_evaluate(&K, sqrtf(dist_sq)); /* :1089-1090 */
                    float *g = &levelset[(i * np + j) * np + k];
                    if (k < upper_k_aligned)
                        *g = fmaf(w, v_i, *g); /* :1101-1107 */
                    else
                        *g += w * v_i;         /* remainder lanes, :1111-1124: mul, then scalar += */
                }
            }
        }
    }
}
#endif

/* ------------------------------------------------------------------------------------------
 * Per-subdomain marching cubes (dense_subdomains.rs:1260-1329, 1470-1578)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    real *vertices;      /* 3 per vertex */
    uint64_t *vertex_keys;
    uint8_t *vertex_interior;
    size_t n_vertices, cap_vertices;
    uint32_t *triangles;  /* 3 per tri, local vertex ids */
    uint8_t *triangle_interior;
    size_t n_triangles, cap_triangles;
    size_t vertex_inside_count, triangle_inside_count;
} patch_t;

static void patch_free(patch_t *p) {
    free(p->vertices);
    free(p->vertex_keys);
    free(p->vertex_interior);
    free(p->triangles);
    free(p->triangle_interior);
    memset(p, 0, sizeof(*p));
}

static void patch_push_vertex(patch_t *p, const real v[3], uint64_t key, int interior) {
    if (p->n_vertices == p->cap_vertices) {
        size_t c = p->cap_vertices ? p->cap_vertices * 2 : 1024;
        p->vertices = (real *)realloc(p->vertices, sizeof(real) * 3 * c);
        p->vertex_keys = (uint64_t *)realloc(p->vertex_keys, sizeof(uint64_t) * c);
        p->vertex_interior = (uint8_t *)realloc(p->vertex_interior, c);
        p->cap_vertices = c;
    }
    memcpy(p->vertices + 3 * p->n_vertices, v, sizeof(real) * 3);
    p->vertex_keys[p->n_vertices] = key;
    p->vertex_interior[p->n_vertices] = (uint8_t)interior;
    p->n_vertices++;
    p->vertex_inside_count += (size_t)interior;
}

static void patch_push_triangle(patch_t *p, const uint32_t t[3], int interior) {
    if (p->n_triangles == p->cap_triangles) {
        size_t c = p->cap_triangles ? p->cap_triangles * 2 : 2048;
        p->triangles = (uint32_t *)realloc(p->triangles, sizeof(uint32_t) * 3 * c);
        p->triangle_interior = (uint8_t *)realloc(p->triangle_interior, c);
        p->cap_triangles = c;
    }
    memcpy(p->triangles + 3 * p->n_triangles, t, sizeof(uint32_t) * 3);
    p->triangle_interior[p->n_triangles] = (uint8_t)interior;
    p->n_triangles++;
    p->triangle_inside_count += (size_t)interior;
}

static void triangulate_subdomain(const sd_params *S, const int64_t sub[3], workspace_t *w, patch_t *patch) {
    const int64_t n = S->subdomain_cubes, np = n + 1;
    const real *G = w->levelset;
    const real t = S->threshold;
    real amin[3], amax[3];
    subdomain_aabb(S, sub, amin, amax);
    SOT(grid) mc;
    int64_t nc3[3] = {n, n, n};
    grid_new(&mc, amin, nc3, S->cube_size);
    const SOT(grid) *gg = &S->global_mc_grid;
    const uint64_t NPy = (uint64_t)gg->n_points[1], NPz = (uint64_t)gg->n_points[2];
    w->n_touched = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = 0; k < n; ++k) { /* flat order :1564-1566 */
                int inside[8];
                int any = 0, case_index = 0;
                for (int c = 0; c < 8; ++c) { /* :1473-1485 */
                    int64_t pi = i + CELL_LOCAL_POINT_COORDS[c][0];
                    int64_t pj = j + CELL_LOCAL_POINT_COORDS[c][1];
                    int64_t pk = k + CELL_LOCAL_POINT_COORDS[c][2];
                    real v = G[(pi * np + pj) * np + pk];
                    inside[c] = v > t;
                    any |= inside[c];
                    case_index |= inside[c] << c; /* marching_cubes_lut.rs:322-329 */
                }
                if (!any) continue;
                const int8_t *row = MC_TABLE[case_index];
                for (int tri = 0; tri < 5 && row[3 * tri] >= 0; ++tri) {
                    uint32_t gt[3];
                    int all_interior = 1;
                    for (int v = 0; v < 3; ++v) {
                        int e = row[3 * tri + v];
                        int oc = CELL_LOCAL_EDGES[e][0], axis = CELL_LOCAL_EDGES[e][1];
                        int64_t o[3] = {i + CELL_LOCAL_POINT_COORDS[oc][0], j + CELL_LOCAL_POINT_COORDS[oc][1],
                                        k + CELL_LOCAL_POINT_COORDS[oc][2]};
                        size_t flat_o = (size_t)((o[0] * np + o[1]) * np + o[2]);
                        size_t slot = flat_o * 3 + (size_t)axis;
                        int32_t vid = w->edge_to_vertex[slot];
                        if (vid < 0) { /* :1498-1539 */
                            int64_t tg[3] = {o[0], o[1], o[2]};
                            tg[axis] += 1;
                            size_t flat_t = (size_t)((tg[0] * np + tg[1]) * np + tg[2]);
                            real ov = G[flat_o], tv = G[flat_t];
                            real alpha = (t - ov) / (tv - ov); /* :1516-1517 */
                            real vc[3];
                            for (int d = 0; d < 3; ++d) {
                                real oc_ = grid_point_coord(&mc, o[d], d);
                                real tc_ = grid_point_coord(&mc, tg[d], d);
                                vc[d] = oc_ * (RC(1.0) - alpha) + tc_ * alpha; /* :1518-1519 */
                            }
                            /* uniform_grid.rs:332-338 */
                            int boundary = 0;
                            for (int d = 0; d < 3; ++d)
                                if (d != axis && (o[d] == 0 || o[d] + 1 == np)) boundary = 1;
                            uint64_t gi = (uint64_t)(sub[0] * n + o[0]), gj = (uint64_t)(sub[1] * n + o[1]),
                                     gk = (uint64_t)(sub[2] * n + o[2]);
                            uint64_t key = ((gi * NPy + gj) * NPz + gk) * 3u + (uint64_t)axis;
                            vid = (int32_t)patch->n_vertices;
                            patch_push_vertex(patch, vc, key, !boundary);
                            w->edge_to_vertex[slot] = vid;
                            if (w->n_touched == w->cap_touched) {
                                size_t c = w->cap_touched ? w->cap_touched * 2 : 4096;
                                w->touched = (uint32_t *)realloc(w->touched, sizeof(uint32_t) * c);
                                w->cap_touched = c;
                            }
                            w->touched[w->n_touched++] = (uint32_t)slot;
                        }
                        gt[v] = (uint32_t)vid;
                        all_interior &= patch->vertex_interior[vid];
                    }
                    patch_push_triangle(patch, gt, all_interior); /* :1544-1551 */
                }
            }
    for (size_t q = 0; q < w->n_touched; ++q) w->edge_to_vertex[w->touched[q]] = -1;
}

static void ws_prepare_levelset(workspace_t *w, int64_t np) {
    size_t tot = (size_t)(np * np * np);
    if (!w->levelset) {
        w->levelset = (real *)malloc(sizeof(real) * tot);
        w->edge_to_vertex = (int32_t *)malloc(sizeof(int32_t) * 3 * tot);
        for (size_t q = 0; q < 3 * tot; ++q) w->edge_to_vertex[q] = -1;
    }
    memset(w->levelset, 0, sizeof(real) * tot); /* :1390-1391 */
}

static void gather_positions_densities(const real *xyz, const real *rho, const uint32_t *idx, size_t P,
                                       workspace_t *w) {
    ws_reserve_particles(w, P);
    for (size_t a = 0; a < P; ++a) {
        const real *p = xyz + 3 * (size_t)idx[a];
        w->pos[3 * a] = p[0];
        w->pos[3 * a + 1] = p[1];
        w->pos[3 * a + 2] = p[2];
        w->rho[a] = rho[idx[a]];
    }
}

/* ------------------------------------------------------------------------------------------
 * Stitching (dense_subdomains.rs:1603-1749); exterior-vertex dedup keyed by the global edge key,
 * which is in 1:1 correspondence with the reference's globalised (subdomain, EdgeIndex) pair
 * (dense_subdomains.rs:1260-1329).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint64_t *keys;
    uint64_t *vals;
    size_t cap, n;
} u64map;

static void map_init(u64map *m, size_t expected) {
    size_t c = 1024;
    while (c < expected * 2 + 16) c <<= 1;
    m->cap = c;
    m->n = 0;
    m->keys = (uint64_t *)malloc(sizeof(uint64_t) * c);
    m->vals = (uint64_t *)malloc(sizeof(uint64_t) * c);
    memset(m->keys, 0xff, sizeof(uint64_t) * c);
}

static uint64_t *map_find_or_insert(u64map *m, uint64_t key, int *inserted) {
    uint64_t hsh = key * 0x9E3779B97F4A7C15ull;
    size_t mask = m->cap - 1, pos = (size_t)(hsh >> 20) & mask;
    while (1) {
        if (m->keys[pos] == key) {
            *inserted = 0;
            return &m->vals[pos];
        }
        if (m->keys[pos] == UINT64_MAX) {
            m->keys[pos] = key;
            m->n++;
            *inserted = 1;
            return &m->vals[pos];
        }
        pos = (pos + 1) & mask;
    }
}

static int stitching(patch_t *patches, int64_t n_patches, SOT(result) *out) {
    size_t total_iv = 0, total_it = 0, total_ev = 0, total_et = 0;
    for (int64_t s = 0; s < n_patches; ++s) {
        total_iv += patches[s].vertex_inside_count;
        total_it += patches[s].triangle_inside_count;
        total_ev += patches[s].n_vertices - patches[s].vertex_inside_count;
        total_et += patches[s].n_triangles - patches[s].triangle_inside_count;
    }
    size_t vcap = total_iv + total_ev, tcap = total_it + total_et;
    real *V = (real *)malloc(sizeof(real) * 3 * (vcap ? vcap : 1));
    uint64_t *VK = (uint64_t *)malloc(sizeof(uint64_t) * (vcap ? vcap : 1));
    uint64_t *T = (uint64_t *)malloc(sizeof(uint64_t) * 3 * (tcap ? tcap : 1));
    u64map map;
    map_init(&map, total_ev);
    size_t v_off = 0, t_off = 0, ext_v = 0, ext_t = 0;
    for (int64_t s = 0; s < n_patches; ++s) {
        patch_t *p = &patches[s];
        uint64_t *l2g = (uint64_t *)malloc(sizeof(uint64_t) * (p->n_vertices ? p->n_vertices : 1));
        size_t new_local = 0;
        for (size_t v = 0; v < p->n_vertices; ++v) {
            if (p->vertex_interior[v]) { /* :1652-1669 */
                size_t g = v_off + new_local++;
                memcpy(V + 3 * g, p->vertices + 3 * v, sizeof(real) * 3);
                VK[g] = p->vertex_keys[v];
                l2g[v] = g;
            } else { /* :1693-1718, first patch wins */
                int inserted;
                uint64_t *slot = map_find_or_insert(&map, p->vertex_keys[v], &inserted);
                if (inserted) {
                    size_t g = total_iv + ext_v++;
                    memcpy(V + 3 * g, p->vertices + 3 * v, sizeof(real) * 3);
                    VK[g] = p->vertex_keys[v];
                    *slot = g;
                }
                l2g[v] = *slot;
            }
        }
        size_t tl = 0;
        for (size_t q = 0; q < p->n_triangles; ++q) {
            const uint32_t *tri = p->triangles + 3 * q;
            size_t g;
            if (p->triangle_interior[q])
                g = t_off + tl++; /* :1671-1691 */
            else
                g = total_it + ext_t++; /* :1720-1733 */
            T[3 * g] = l2g[tri[0]];
            T[3 * g + 1] = l2g[tri[1]];
            T[3 * g + 2] = l2g[tri[2]];
        }
        v_off += p->vertex_inside_count;
        t_off += p->triangle_inside_count;
        free(l2g);
    }
    free(map.keys);
    free(map.vals);
    out->n_vertices = total_iv + ext_v;
    out->vertices = V;
    out->vertex_keys = VK;
    out->n_triangles = total_it + ext_t;
    out->triangles = T;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Entry points (lib.rs:330-473, reconstruction.rs:17-62)
 * ------------------------------------------------------------------------------------------ */
/* ------------------------------------------------------------------------------------------
 * Global (non-decomposed) strategy, SURVEY rows A14/A15: reconstruction.rs:65-194 with
 * enable_multi_threading = false, i.e. the reference's *sequential* functions, which are the only
 * deterministic ones of this strategy (the parallel variants fill dashmaps / merge thread-local maps in
 * thread-timing order).  Hash maps of the reference (cell -> particles, point -> value, cell -> CellData)
 * are replaced by dense arrays over the grid; missing map entries and zeros are interchangeable in every
 * test the reference performs on them (narrow_band_extraction.rs:79-88, 161-176).
 * ------------------------------------------------------------------------------------------ */
static int global_densities_and_neighbors(const SOT(grid) *grid, const real *xyz, uint64_t n, const SOT(params) *P, real mass,
                                          int nthreads, real *rho, uint64_t **out_ptr, uint64_t **out_nb) {
    /* neighborhood_search.rs:148-230 (sequential spatial hashing on domain = grid.aabb(), cell size h)
       + density_map.rs:113-186 */
    const real h = P->compact_support_radius;
    if (!(h > RC(0.0))) return 4; /* assert, neighborhood_search.rs:159-162 */
    SOT(grid) sgrid;
    if (grid_from_aabb(&sgrid, grid->aabb_min, grid->aabb_max, h) != 0) return 4; /* asserts :163-170, expect :176 */
    size_t ncell = (size_t)(sgrid.n_cells[0] * sgrid.n_cells[1] * sgrid.n_cells[2]);
    uint32_t *cell_start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    uint32_t *cell_of = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *items = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint64_t a = 0; a < n; ++a) { /* neighborhood_search.rs:655-676: ascending particle index per cell */
        int64_t c[3];
        grid_enclosing_cell(&sgrid, xyz + 3 * a, c);
        if (!grid_cell_exists(&sgrid, c)) { /* get_cell(..).unwrap() */
            free(cell_start); free(cell_of); free(items);
            return 4;
        }
        cell_of[a] = (uint32_t)grid_flatten_cell(&sgrid, c);
        cell_start[cell_of[a] + 1]++;
    }
    for (size_t c = 0; c < ncell; ++c) cell_start[c + 1] += cell_start[c];
    {
        uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * (ncell ? ncell : 1));
        memcpy(cursor, cell_start, sizeof(uint32_t) * ncell);
        for (uint64_t a = 0; a < n; ++a) items[cursor[cell_of[a]]++] = (uint32_t)a;
        free(cursor);
    }
    cubic_kernel K = kernel_new(h);
    const real h2 = h * h;
    const real w0 = kernel_evaluate(&K, RC(0.0));
    uint64_t *ptr = (uint64_t *)calloc((size_t)n + 1, sizeof(uint64_t));
    for (int pass = 0; pass < 2; ++pass) { /* pass 0: counts + densities, pass 1: lists */
        uint64_t *nb = NULL;
        if (pass == 1) {
            uint64_t run = 0;
            for (uint64_t a = 0; a < n; ++a) {
                uint64_t c = ptr[a];
                ptr[a] = run;
                run += c;
            }
            ptr[n] = run;
            nb = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(run ? run : 1));
            *out_nb = nb;
        }
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t a = 0; a < (int64_t)n; ++a) {
            const real *pi = xyz + 3 * a;
            int64_t ci[3];
            grid_enclosing_cell(&sgrid, pi, ci);
            real density = w0;
            uint64_t cnt = 0;
            /* cells_adjacent_to_cell (26, iproduct order) chained with the cell itself (:189-197) */
            for (int ph = 0; ph < 2; ++ph)
                for (int sx = -1; sx <= 1; ++sx)
                    for (int sy = -1; sy <= 1; ++sy)
                        for (int sz = -1; sz <= 1; ++sz) {
                            int is_center = (sx == 0 && sy == 0 && sz == 0);
                            if ((ph == 0) == is_center) continue;
                            int64_t c[3] = {ci[0] + sx, ci[1] + sy, ci[2] + sz};
                            if (!grid_cell_exists(&sgrid, c)) continue;
                            size_t f = (size_t)grid_flatten_cell(&sgrid, c);
                            for (uint32_t q = cell_start[f]; q < cell_start[f + 1]; ++q) {
                                uint32_t b = items[q];
                                if ((int64_t)b == a) continue; /* :216-218 */
                                const real *pj = xyz + 3 * (size_t)b;
                                real dx = pj[0] - pi[0], dy = pj[1] - pi[1], dz = pj[2] - pi[2];
                                real d2 = dx * dx + dy * dy + dz * dz;
                                if (d2 < h2) { /* :221 */
                                    if (pass == 0)
                                        density += kernel_evaluate(&K, R_SQRT(d2)); /* density_map.rs:176-180 */
                                    else
                                        nb[ptr[a] + cnt] = b;
                                    ++cnt;
                                }
                            }
                        }
            if (pass == 0) {
                rho[a] = density * mass; /* density_map.rs:182 */
                ptr[a] = cnt;
            }
        }
    }
    *out_ptr = ptr;
    free(cell_start);
    free(cell_of);
    free(items);
    return 0;
}

/* density_map.rs:364-412, 582-737 (SparseDensityMapGenerator, sequential) into a dense array */
static int global_density_map(const SOT(grid) *grid, const real *xyz, const real *rho, uint64_t n, const SOT(params) *P, real mass,
                              real *levelset) {
    const real h = P->compact_support_radius, cs = P->cube_size;
    /* compute_kernel_evaluation_radius, density_map.rs:551-580 */
    const real half_supported_cells_real = R_CEIL(h / cs);
    const int64_t half_supported_cells = (int64_t)(double)half_supported_cells_real;
    const int64_t supported_points = 1 + (half_supported_cells * 2 + 1);
    const real radius = cs * half_supported_cells_real * (RC(1.0) + R_SQRT(R_EPSILON));
    const real radius_sq = radius * radius;
    cubic_kernel K = kernel_new(h);
    real amin[3], amax[3]; /* allowed domain :606-613 */
    const real neg = -radius;
    for (int d = 0; d < 3; ++d) {
        amin[d] = grid->aabb_min[d] - neg;
        amax[d] = grid->aabb_max[d] + neg;
    }
    if ((amin[0] == amax[0] && amin[1] == amax[1] && amin[2] == amax[2]) ||
        !(amin[0] <= amax[0] && amin[1] <= amax[1] && amin[2] <= amax[2]))
        return 2; /* DensityMapError::InvalidDomain :615-627 */
    const int64_t npy = grid->n_points[1], npz = grid->n_points[2];
    for (uint64_t a = 0; a < n; ++a) { /* ascending particle order :389-396 */
        const real *p = xyz + 3 * a;
        if (!(p[0] >= amin[0] && p[1] >= amin[1] && p[2] >= amin[2] && p[0] < amax[0] && p[1] < amax[1] && p[2] < amax[2]))
            continue; /* :648-651 */
        int64_t cell[3], lo[3], hi[3];
        grid_enclosing_cell(grid, p, cell);
        for (int d = 0; d < 3; ++d) {
            lo[d] = cell[d] - half_supported_cells;
            hi[d] = lo[d] + supported_points;
        }
        const real volume = mass / rho[a]; /* :688 */
        real mp[3];
        for (int d = 0; d < 3; ++d) mp[d] = grid_point_coord(grid, lo[d], d);
        real dx = mp[0] - p[0] - cs; /* :694-697 */
        for (int64_t i = lo[0]; i != hi[0]; ++i) {
            dx += cs;
            const real dxdx = dx * dx;
            real dy = mp[1] - p[1] - cs;
            for (int64_t j = lo[1]; j != hi[1]; ++j) {
                dy += cs;
                const real dydy = dy * dy;
                real dz = mp[2] - p[2] - cs;
                for (int64_t k = lo[2]; k != hi[2]; ++k) {
                    dz += cs;
                    const real dzdz = dz * dz;
                    const real r2 = dxdx + dydy + dzdz;
                    if (r2 < radius_sq) {
                        const real contribution = volume * kernel_evaluate(&K, R_SQRT(r2));
                        levelset[(i * npy + j) * npz + k] += contribution; /* :722-726 */
                    }
                }
            }
        }
    }
    return 0;
}

/* narrow_band_extraction.rs:8-219 + triangulation.rs:23-95.  Output order is canonical (the reference's
   is hash-map order): vertices by ascending edge key, triangles by ascending flat cell index. */
static int global_marching_cubes(const SOT(grid) *grid, const real *G, real t, SOT(result) *out) {
    const int64_t np[3] = {grid->n_points[0], grid->n_points[1], grid->n_points[2]};
    const int64_t nc[3] = {grid->n_cells[0], grid->n_cells[1], grid->n_cells[2]};
    const size_t npts = (size_t)(np[0] * np[1] * np[2]);
    /* vertex id of the edge starting at a point in +axis direction, -1 = no iso-surface vertex */
    int64_t *edge_vertex = (int64_t *)malloc(sizeof(int64_t) * 3 * (npts ? npts : 1));
    uint64_t nv = 0;
    for (size_t q = 0; q < 3 * npts; ++q) edge_vertex[q] = -1;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            out->n_vertices = nv;
            out->vertices = (real *)malloc(sizeof(real) * 3 * (size_t)(nv ? nv : 1));
            out->vertex_keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nv ? nv : 1));
            nv = 0;
        }
        for (int64_t i = 0; i < np[0]; ++i)
            for (int64_t j = 0; j < np[1]; ++j)
                for (int64_t k = 0; k < np[2]; ++k) {
                    const int64_t o[3] = {i, j, k};
                    const size_t fo = (size_t)((i * np[1] + j) * np[2] + k);
                    for (int axis = 0; axis < 3; ++axis) {
                        if (o[axis] + 1 >= np[axis]) continue; /* get_point_neighbor: no neighbour outside the grid */
                        int64_t q[3] = {i, j, k};
                        q[axis] += 1;
                        const size_t fq = (size_t)((q[0] * np[1] + q[1]) * np[2] + q[2]);
                        const real vo = G[fo], vq = G[fq];
                        /* the edge is visited from its endpoint with value >= t towards a neighbour < t (:69-92) */
                        int from_o = !(vo < t) && (vq < t);
                        int from_q = !(vq < t) && (vo < t);
                        if (!from_o && !from_q) continue;
                        if (pass == 0) {
                            ++nv;
                            continue;
                        }
                        const int64_t *pp = from_o ? o : q, *nn = from_o ? q : o;
                        const real pv = from_o ? vo : vq, nvv = from_o ? vq : vo;
                        const real alpha = (t - pv) / (nvv - pv); /* :95 */
                        for (int d = 0; d < 3; ++d) {
                            const real pc = grid_point_coord(grid, pp[d], d), ncd = grid_point_coord(grid, nn[d], d);
                            out->vertices[3 * nv + d] = pc * (RC(1.0) - alpha) + ncd * alpha; /* :96-99 */
                        }
                        out->vertex_keys[nv] = (uint64_t)fo * 3u + (uint64_t)axis;
                        edge_vertex[3 * fo + (size_t)axis] = (int64_t)nv;
                        ++nv;
                    }
                }
    }
    /* cells: every existing cell adjacent to an edge with a vertex (:107-129); corner flags (:115-127, 161-176) */
    uint64_t nt = 0, cap = 1024;
    uint64_t *tris = (uint64_t *)malloc(sizeof(uint64_t) * 3 * cap);
    int rc = 0;
    for (int64_t i = 0; i < nc[0] && !rc; ++i)
        for (int64_t j = 0; j < nc[1] && !rc; ++j)
            for (int64_t k = 0; k < nc[2] && !rc; ++k) {
                int64_t ev[12];
                int any = 0;
                for (int e = 0; e < 12; ++e) {
                    const int oc = CELL_LOCAL_EDGES[e][0], axis = CELL_LOCAL_EDGES[e][1];
                    const int64_t o[3] = {i + CELL_LOCAL_POINT_COORDS[oc][0], j + CELL_LOCAL_POINT_COORDS[oc][1],
                                          k + CELL_LOCAL_POINT_COORDS[oc][2]};
                    ev[e] = edge_vertex[3 * (size_t)((o[0] * np[1] + o[1]) * np[2] + o[2]) + (size_t)axis];
                    any |= ev[e] >= 0;
                }
                if (!any) continue;
                int case_index = 0;
                for (int c = 0; c < 8; ++c) {
                    const int64_t pc[3] = {i + CELL_LOCAL_POINT_COORDS[c][0], j + CELL_LOCAL_POINT_COORDS[c][1],
                                           k + CELL_LOCAL_POINT_COORDS[c][2]};
                    const real v = G[(pc[0] * np[1] + pc[1]) * np[2] + pc[2]];
                    int above = v > t; /* :161-170 */
                    if (!above && !(v < t)) {
                        /* v == t: marked Above only through a crossing edge of this cell that starts at it (:115-127) */
                        for (int e = 0; e < 12 && !above; ++e) {
                            if (ev[e] < 0) continue;
                            const int oc = CELL_LOCAL_EDGES[e][0], axis = CELL_LOCAL_EDGES[e][1];
                            int tc[3] = {CELL_LOCAL_POINT_COORDS[oc][0], CELL_LOCAL_POINT_COORDS[oc][1], CELL_LOCAL_POINT_COORDS[oc][2]};
                            int is_o = tc[0] == CELL_LOCAL_POINT_COORDS[c][0] && tc[1] == CELL_LOCAL_POINT_COORDS[c][1] &&
                                       tc[2] == CELL_LOCAL_POINT_COORDS[c][2];
                            tc[axis] += 1;
                            int is_t = tc[0] == CELL_LOCAL_POINT_COORDS[c][0] && tc[1] == CELL_LOCAL_POINT_COORDS[c][1] &&
                                       tc[2] == CELL_LOCAL_POINT_COORDS[c][2];
                            if (is_o || is_t) above = 1; /* the other endpoint is < t, so this corner is the >= t one */
                        }
                    }
                    case_index |= above << c;
                }
                const int8_t *row = MC_TABLE[case_index];
                for (int tri = 0; tri < 5 && row[3 * tri] >= 0; ++tri) {
                    if (nt == cap) {
                        cap *= 2;
                        tris = (uint64_t *)realloc(tris, sizeof(uint64_t) * 3 * cap);
                    }
                    for (int v = 0; v < 3; ++v) {
                        const int64_t id = ev[row[3 * tri + v]];
                        if (id < 0) rc = 3; /* "Missing iso surface vertex", triangulation.rs:62-95 */
                        tris[3 * nt + (size_t)v] = (uint64_t)id;
                    }
                    ++nt;
                }
            }
    free(edge_vertex);
    out->n_triangles = nt;
    out->triangles = tris;
    return rc;
}

static int reconstruct_surface_global(const real *xyz, uint64_t n, const SOT(params) *P, const SOT(grid) *grid, int nthreads,
                                      SOT(result) *out) {
    /* reconstruction.rs:65-194 */
    const real d = P->particle_radius + P->particle_radius; /* kernel.rs:28-30 Volume::cube_particle */
    const real mass = d * d * d * P->rest_density;
    out->grid = *grid;
    memset(&out->subdomain_grid, 0, sizeof(out->subdomain_grid));
    double t1 = now_s();
    real *rho = (real *)calloc(n ? n : 1, sizeof(real));
    out->particle_densities = rho;
    int rc = global_densities_and_neighbors(grid, xyz, n, P, mass, nthreads, rho, &out->neighbor_ptr, &out->neighbors);
    if (rc) return rc;
    double t2 = now_s();
    const size_t npts = (size_t)(grid->n_points[0] * grid->n_points[1] * grid->n_points[2]);
    real *G = (real *)calloc(npts ? npts : 1, sizeof(real));
    rc = global_density_map(grid, xyz, rho, n, P, mass, G);
    if (!rc) rc = global_marching_cubes(grid, G, P->iso_surface_threshold, out);
    out->global_levelset = G;
    double t3 = now_s();
    out->t_density = t2 - t1;
    out->t_reconstruction = t3 - t2;
    return rc;
}

/* stand-alone marching cubes on a dense array (marching_cubes.rs:100-127 with DensityMap::Dense; pysplashsurf.marching_cubes):
   grid = UniformGrid::new(translation, n_points - 1, cube_size).  Returns 0, 1 (grid error) or 3 (triangulation error). */
int SOFN(marching_cubes)(const real *values, const int64_t n_points[3], real threshold, real cube_size, const real translation[3], SOT(result) *out) {
    memset(out, 0, sizeof(*out));
    if (!(cube_size > RC(0.0))) return 1;
    int64_t nc[3];
    for (int d = 0; d < 3; ++d) {
        if (n_points[d] < 2) return 1;
        nc[d] = n_points[d] - 1;
    }
    SOT(grid) g;
    grid_new(&g, translation, nc, cube_size);
    out->grid = g;
    out->used_global_strategy = 1;
    return global_marching_cubes(&g, values, threshold, out);
}

static int resolve_threads(const SOT(params) *P) {
#ifdef _OPENMP
    int t = P->num_threads > 0 ? P->num_threads : omp_get_max_threads();
    return t < 1 ? 1 : t;
#else
    (void)P;
    return 1;
#endif
}

int SOFN(grid_for_reconstruction)(const real *xyz, uint64_t n, const SOT(params) *P, SOT(grid) *out) {
    if (P->has_particle_aabb) {
        /* grid only depends on the AABB in this case */
        return grid_for_reconstruction(xyz, n, P, out) ? 1 : 0;
    }
    return grid_for_reconstruction(xyz, n, P, out) ? 1 : 0;
}

static real *filter_particles(const real *xyz, uint64_t n, const SOT(params) *P, SOT(result) *out, uint64_t *n_out) {
    /* lib.rs:369-406 */
    if (!P->has_particle_aabb) {
        out->particle_inside_aabb = NULL;
        *n_out = n;
        return NULL;
    }
    uint8_t *inside = (uint8_t *)malloc(n ? n : 1);
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const real *p = xyz + 3 * i;
        int in = p[0] >= P->aabb_min[0] && p[1] >= P->aabb_min[1] && p[2] >= P->aabb_min[2] &&
                 p[0] < P->aabb_max[0] && p[1] < P->aabb_max[1] && p[2] < P->aabb_max[2];
        inside[i] = (uint8_t)in;
        cnt += (uint64_t)in;
    }
    real *f = (real *)malloc(sizeof(real) * 3 * (cnt ? cnt : 1));
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (inside[i]) {
            memcpy(f + 3 * k, xyz + 3 * i, sizeof(real) * 3);
            ++k;
        }
    out->particle_inside_aabb = inside;
    *n_out = cnt;
    return f;
}

int SOFN(reconstruct_surface)(const real *xyz_in, uint64_t n_in, const SOT(params) *P, SOT(result) *out) {
    memset(out, 0, sizeof(*out));
    if (!(P->cube_size > RC(0.0)) || !(P->compact_support_radius >= RC(0.0))) return 4; /* reference panics */
    double t0 = now_s();
    int nthreads = resolve_threads(P);
    out->threads_used = nthreads;
    out->n_input = n_in;
    uint64_t n = 0;
    real *filtered = filter_particles(xyz_in, n_in, P, out, &n);
    const real *xyz = filtered ? filtered : xyz_in;
    out->n_particles = n;

    SOT(grid) initial;
    if (grid_for_reconstruction(xyz, n, P, &initial) != 0) {
        free(filtered);
        return 1;
    }
    /* strategy choice, lib.rs:419-462 */
    int use_decomposition = 1;
    if (P->global_strategy == 1) {
        use_decomposition = 0; /* SpatialDecomposition::None */
    } else if (P->global_strategy == 2) { /* UniformGrid with auto_disable */
        int64_t max_cubes = initial.n_cells[0];
        if (initial.n_cells[1] > max_cubes) max_cubes = initial.n_cells[1];
        if (initial.n_cells[2] > max_cubes) max_cubes = initial.n_cells[2];
        uint32_t with_margin = (uint32_t)(1.2 * (double)P->subdomain_num_cubes_per_dim);
        uint32_t mc32 = max_cubes > (int64_t)UINT32_MAX ? UINT32_MAX : (uint32_t)max_cubes;
        use_decomposition = mc32 > with_margin;
    }
    if (!use_decomposition) {
        int rc = reconstruct_surface_global(xyz, n, P, &initial, nthreads, out);
        free(filtered);
        out->used_global_strategy = 1;
        out->t_total = now_s() - t0;
        return rc;
    }
    sd_params S;
    initialize_parameters(P, &initial, &S);
    out->grid = S.global_mc_grid;
    out->subdomain_grid = S.subdomain_grid;

    double t1 = now_s();
    subdomains_t subs;
    memset(&subs, 0, sizeof(subs));
    if (decomposition(&S, xyz, n, &subs, nthreads) != 0) {
        free(filtered);
        return 4;
    }
    out->n_subdomains = subs.n_sub;
    out->n_subdomain_particles = subs.offsets[subs.n_sub];
    double t2 = now_s();

    real *rho = (real *)calloc(n ? n : 1, sizeof(real)); /* :504 */
    uint32_t **nb_lists = NULL;
    uint32_t *nb_counts = NULL;
    if (P->global_neighborhood_list) { /* :505 */
        nb_lists = (uint32_t **)calloc(n ? n : 1, sizeof(uint32_t *));
        nb_counts = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    }
    workspace_t *ws = (workspace_t *)calloc((size_t)nthreads, sizeof(workspace_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t s = 0; s < subs.n_sub; ++s) {
#ifdef _OPENMP
        workspace_t *w = &ws[omp_get_thread_num()];
#else
        workspace_t *w = &ws[0];
#endif
        subdomain_density(&S, xyz, subs.particles + subs.offsets[s], (size_t)(subs.offsets[s + 1] - subs.offsets[s]),
                          subs.flat_index[s], w, rho, nb_lists, nb_counts);
    }
    if (nb_lists) {
        out->neighbor_ptr = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n + 1));
        uint64_t run = 0;
        for (uint64_t i = 0; i < n; ++i) {
            out->neighbor_ptr[i] = run;
            run += nb_counts[i];
        }
        out->neighbor_ptr[n] = run;
        out->neighbors = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(run ? run : 1));
        for (uint64_t i = 0; i < n; ++i) {
            for (uint32_t q = 0; q < nb_counts[i]; ++q) out->neighbors[out->neighbor_ptr[i] + q] = nb_lists[i][q];
            free(nb_lists[i]);
        }
        free(nb_lists);
        free(nb_counts);
    }
    double t3 = now_s();

    patch_t *patches = (patch_t *)calloc((size_t)(subs.n_sub ? subs.n_sub : 1), sizeof(patch_t));
    const int64_t np = S.subdomain_cubes + 1;
    size_t max_particles = 0; /* dense_subdomains.rs:1241-1253 */
    for (int64_t s = 0; s < subs.n_sub; ++s) {
        size_t P_s = (size_t)(subs.offsets[s + 1] - subs.offsets[s]);
        if (P_s > max_particles) max_particles = P_s;
    }
    size_t sparse_limit = max_particles / (100 / 5);
    if (sparse_limit < 100) sparse_limit = 100;
    (void)sparse_limit;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t s = 0; s < subs.n_sub; ++s) {
#ifdef _OPENMP
        workspace_t *w = &ws[omp_get_thread_num()];
#else
        workspace_t *w = &ws[0];
#endif
        size_t P_s = (size_t)(subs.offsets[s + 1] - subs.offsets[s]);
        gather_positions_densities(xyz, rho, subs.particles + subs.offsets[s], P_s, w);
        ws_prepare_levelset(w, np);
        int64_t sub[3];
        grid_unflatten_cell(&S.subdomain_grid, subs.flat_index[s], sub);
#ifndef SO_F64
        if (P->enable_simd == 2 || (P->enable_simd == 1 && P_s > sparse_limit)) /* :1590-1596, :1413-1415 */
            density_grid_loop_avx(&S, sub, w->pos, w->rho, P_s, w->levelset, P->enable_simd == 2);
        else
#endif
            density_grid_loop_scalar(&S, sub, w->pos, w->rho, P_s, w->levelset);
        triangulate_subdomain(&S, sub, w, &patches[s]);
    }
    double t4 = now_s();

    stitching(patches, subs.n_sub, out);
    double t5 = now_s();

    for (int64_t s = 0; s < subs.n_sub; ++s) patch_free(&patches[s]);
    free(patches);
    for (int t = 0; t < nthreads; ++t) ws_free(&ws[t]);
    free(ws);
    subdomains_free(&subs);
    free(filtered);
    out->particle_densities = rho;
    out->t_decomposition = t2 - t1;
    out->t_density = t3 - t2;
    out->t_reconstruction = t4 - t3;
    out->t_stitching = t5 - t4;
    out->t_total = t5 - t0;
    return 0;
}

int64_t SOFN(debug_levelset_subdomain)(const real *xyz, uint64_t n, const SOT(params) *P, int64_t flat_subdomain,
                                    real *out_grid) {
    SOT(grid) initial;
    if (P->has_particle_aabb) return -2; /* not supported by this debug entry */
    if (grid_for_reconstruction(xyz, n, P, &initial) != 0) return -2;
    sd_params S;
    initialize_parameters(P, &initial, &S);
    int nthreads = resolve_threads(P);
    subdomains_t subs;
    memset(&subs, 0, sizeof(subs));
    if (decomposition(&S, xyz, n, &subs, nthreads) != 0) return -2;
    real *rho = (real *)calloc(n ? n : 1, sizeof(real));
    workspace_t *ws = (workspace_t *)calloc((size_t)nthreads, sizeof(workspace_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t s = 0; s < subs.n_sub; ++s) {
#ifdef _OPENMP
        workspace_t *w = &ws[omp_get_thread_num()];
#else
        workspace_t *w = &ws[0];
#endif
        subdomain_density(&S, xyz, subs.particles + subs.offsets[s], (size_t)(subs.offsets[s + 1] - subs.offsets[s]),
                          subs.flat_index[s], w, rho, NULL, NULL);
    }
    int64_t result = -1;
    const int64_t np = S.subdomain_cubes + 1;
    for (int64_t s = 0; s < subs.n_sub; ++s) {
        if (subs.flat_index[s] != flat_subdomain) continue;
        workspace_t *w = &ws[0];
        size_t P_s = (size_t)(subs.offsets[s + 1] - subs.offsets[s]);
        gather_positions_densities(xyz, rho, subs.particles + subs.offsets[s], P_s, w);
        ws_prepare_levelset(w, np);
        int64_t sub[3];
        grid_unflatten_cell(&S.subdomain_grid, flat_subdomain, sub);
#ifndef SO_F64
        size_t max_particles = 0;
        for (int64_t q = 0; q < subs.n_sub; ++q)
            if ((size_t)(subs.offsets[q + 1] - subs.offsets[q]) > max_particles) max_particles = (size_t)(subs.offsets[q + 1] - subs.offsets[q]);
        size_t sparse_limit = max_particles / (100 / 5);
        if (sparse_limit < 100) sparse_limit = 100;
        if (P->enable_simd == 2 || (P->enable_simd == 1 && P_s > sparse_limit))
            density_grid_loop_avx(&S, sub, w->pos, w->rho, P_s, w->levelset, P->enable_simd == 2);
        else
#endif
            density_grid_loop_scalar(&S, sub, w->pos, w->rho, P_s, w->levelset);
        memcpy(out_grid, w->levelset, sizeof(real) * (size_t)(np * np * np));
        result = (int64_t)P_s;
    }
    for (int t = 0; t < nthreads; ++t) ws_free(&ws[t]);
    free(ws);
    free(rho);
    subdomains_free(&subs);
    return result;
}

/* ------------------------------------------------------------------------------------------
 * Sharded variant (multi-process): same functions, restricted to a box of subdomains of the grid of
 * the whole job.  Mirrors include/splashsurf_hip.h ss_shard_begin_f32 / ss_shard_finish.
 * ------------------------------------------------------------------------------------------ */
static int shard_setup(const SOT(params) *P, const SOT(shard) *sh, sd_params *S) {
    if (P->has_particle_aabb) return 4;
    SOT(grid) initial;
    if (grid_for_particle_aabb(sh->domain_min, sh->domain_max, P, &initial) != 0) return 1;
    initialize_parameters(P, &initial, S);
    for (int d = 0; d < 3; ++d)
        if (sh->sub_lo[d] < 0 || sh->sub_hi[d] > S->subdomain_grid.n_cells[d] || sh->sub_lo[d] > sh->sub_hi[d]) return 4;
    return 0;
}

int SOFN(grid_for_domain)(const SOT(params) *P, const real dmin[3], const real dmax[3], SOT(grid) *grid, SOT(grid) *subgrid, real *margin) {
    SOT(grid) initial;
    if (grid_for_particle_aabb(dmin, dmax, P, &initial) != 0) return 1;
    sd_params S;
    initialize_parameters(P, &initial, &S);
    *grid = S.global_mc_grid;
    *subgrid = S.subdomain_grid;
    if (margin) *margin = S.ghost_margin;
    return 0;
}

int SOFN(shard_densities)(const real *xyz, uint64_t n, const SOT(params) *P, const SOT(shard) *sh, real *rho_out) {
    sd_params S;
    int rc = shard_setup(P, sh, &S);
    if (rc) return rc;
    int nthreads = resolve_threads(P);
    subdomains_t subs;
    memset(&subs, 0, sizeof(subs));
    if (decomposition_boxed(&S, xyz, n, &subs, nthreads, sh->sub_lo, sh->sub_hi) != 0) return 4;
    memset(rho_out, 0, sizeof(real) * (size_t)n);
    workspace_t *ws = (workspace_t *)calloc((size_t)nthreads, sizeof(workspace_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t s = 0; s < subs.n_sub; ++s) {
#ifdef _OPENMP
        workspace_t *w = &ws[omp_get_thread_num()];
#else
        workspace_t *w = &ws[0];
#endif
        subdomain_density(&S, xyz, subs.particles + subs.offsets[s], (size_t)(subs.offsets[s + 1] - subs.offsets[s]), subs.flat_index[s], w,
                          rho_out, NULL, NULL);
    }
    for (int t = 0; t < nthreads; ++t) ws_free(&ws[t]);
    free(ws);
    subdomains_free(&subs);
    return 0;
}

int SOFN(shard_reconstruct)(const real *xyz, uint64_t n, const SOT(params) *P, const SOT(shard) *sh, const real *rho, SOT(result) *out) {
    memset(out, 0, sizeof(*out));
    sd_params S;
    int rc = shard_setup(P, sh, &S);
    if (rc) return rc;
    int nthreads = resolve_threads(P);
    out->threads_used = nthreads;
    out->n_input = out->n_particles = n;
    out->grid = S.global_mc_grid;
    out->subdomain_grid = S.subdomain_grid;
    subdomains_t subs;
    memset(&subs, 0, sizeof(subs));
    if (decomposition_boxed(&S, xyz, n, &subs, nthreads, sh->sub_lo, sh->sub_hi) != 0) return 4;
    out->n_subdomains = subs.n_sub;
    out->n_subdomain_particles = subs.offsets[subs.n_sub];
    workspace_t *ws = (workspace_t *)calloc((size_t)nthreads, sizeof(workspace_t));
    patch_t *patches = (patch_t *)calloc((size_t)(subs.n_sub ? subs.n_sub : 1), sizeof(patch_t));
    const int64_t np = S.subdomain_cubes + 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t s = 0; s < subs.n_sub; ++s) {
#ifdef _OPENMP
        workspace_t *w = &ws[omp_get_thread_num()];
#else
        workspace_t *w = &ws[0];
#endif
        size_t P_s = (size_t)(subs.offsets[s + 1] - subs.offsets[s]);
        gather_positions_densities(xyz, rho, subs.particles + subs.offsets[s], P_s, w);
        ws_prepare_levelset(w, np);
        int64_t sub[3];
        grid_unflatten_cell(&S.subdomain_grid, subs.flat_index[s], sub);
        density_grid_loop_scalar(&S, sub, w->pos, w->rho, P_s, w->levelset);
        triangulate_subdomain(&S, sub, w, &patches[s]);
    }
    stitching(patches, subs.n_sub, out);
    for (int64_t s = 0; s < subs.n_sub; ++s) patch_free(&patches[s]);
    free(patches);
    for (int t = 0; t < nthreads; ++t) ws_free(&ws[t]);
    free(ws);
    subdomains_free(&subs);
    out->particle_densities = (real *)malloc(sizeof(real) * (size_t)(n ? n : 1));
    memcpy(out->particle_densities, rho, sizeof(real) * (size_t)n);
    return 0;
}

int64_t SOFN(debug_shard_levelset)(const real *xyz, uint64_t n, const SOT(params) *P, const SOT(shard) *sh, const real *rho, int64_t flat_subdomain,
                                real *out_grid) {
    sd_params S;
    if (shard_setup(P, sh, &S)) return -2;
    subdomains_t subs;
    memset(&subs, 0, sizeof(subs));
    if (decomposition_boxed(&S, xyz, n, &subs, 1, sh->sub_lo, sh->sub_hi) != 0) return -2;
    int64_t result = -1;
    const int64_t np = S.subdomain_cubes + 1;
    workspace_t w;
    memset(&w, 0, sizeof(w));
    for (int64_t s = 0; s < subs.n_sub; ++s) {
        if (subs.flat_index[s] != flat_subdomain) continue;
        size_t P_s = (size_t)(subs.offsets[s + 1] - subs.offsets[s]);
        gather_positions_densities(xyz, rho, subs.particles + subs.offsets[s], P_s, &w);
        ws_prepare_levelset(&w, np);
        int64_t sub[3];
        grid_unflatten_cell(&S.subdomain_grid, flat_subdomain, sub);
        density_grid_loop_scalar(&S, sub, w.pos, w.rho, P_s, w.levelset);
        memcpy(out_grid, w.levelset, sizeof(real) * (size_t)(np * np * np));
        result = (int64_t)P_s;
    }
    ws_free(&w);
    subdomains_free(&subs);
    return result;
}

void SOFN(result_free)(SOT(result) *r) {
    free(r->particle_densities);
    free(r->particle_inside_aabb);
    free(r->neighbor_ptr);
    free(r->neighbors);
    free(r->vertices);
    free(r->vertex_keys);
    free(r->triangles);
    free(r->global_levelset);
    memset(r, 0, sizeof(*r));
}
