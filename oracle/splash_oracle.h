/* splash_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's subdomain-grid surface reconstruction
 * (splashsurf_lib::reconstruct_surface with SpatialDecomposition::UniformGrid, scalar / non-SIMD code
 * path), following the reference function by function.  Every function in splash_oracle.c cites the
 * reference file:line it restates.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product (libsplashsurf_hip.so) never does.
 *
 * Parity pinning: validated in the build container against the reference's own pre-built wheel
 * (tools/gen_goldens.py; fixtures in tests/golden/): per-particle densities bit-identical, meshes
 * identical up to vertex order (see DESIGN.md "Oracle").
 */
#ifndef SPLASH_ORACLE_H
#define SPLASH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_params {
    float particle_radius;
    float rest_density;
    float compact_support_radius; /* absolute units (lib.rs:163-164) */
    float cube_size;              /* absolute units (lib.rs:165-166) */
    float iso_surface_threshold;
    int32_t has_particle_aabb;
    float aabb_min[3];
    float aabb_max[3];
    int32_t subdomain_num_cubes_per_dim; /* lib.rs:142, default 64 */
    int32_t num_threads;                 /* <=0: all cores (OpenMP) */
    int32_t global_neighborhood_list;    /* lib.rs:185-188 */
} so_params;

typedef struct so_grid {
    float aabb_min[3];
    float aabb_max[3];
    float cell_size;
    int64_t n_points[3];
    int64_t n_cells[3];
} so_grid;

typedef struct so_result {
    so_grid grid;            /* padded global MC grid (reconstruction.rs:27-29) */
    so_grid subdomain_grid;  /* reconstruction.rs:26 */
    uint64_t n_input;        /* input particle count */
    uint64_t n_particles;    /* after the optional AABB filter */
    float *particle_densities;      /* [n_particles] */
    uint8_t *particle_inside_aabb;  /* [n_input] or NULL when no AABB given */
    uint64_t *neighbor_ptr;         /* [n_particles+1] CSR rows or NULL (dense_subdomains.rs:617-639) */
    uint64_t *neighbors;            /* global particle indices */
    uint64_t n_vertices;
    float *vertices;         /* [n_vertices*3] */
    uint64_t *vertex_keys;   /* [n_vertices] global edge key = ((gi*NPy+gj)*NPz+gk)*3+axis */
    uint64_t n_triangles;
    uint64_t *triangles;     /* [n_triangles*3] */
    int64_t n_subdomains;    /* occupied subdomains (patches) */
    uint64_t n_subdomain_particles; /* sum over subdomains incl. ghosts */
    double t_total, t_decomposition, t_density, t_reconstruction, t_stitching; /* seconds */
    int32_t threads_used;
} so_result;

/* returns 0 on success; 1 grid construction error; 4 other */
int so_reconstruct_surface(const float *xyz, uint64_t n, const so_params *params, so_result *out);
void so_result_free(so_result *r);

/* lib.rs:476-516 */
int so_grid_for_reconstruction(const float *xyz, uint64_t n, const so_params *params, so_grid *out);

/* Debug/observability entry points used by the parity tests */
/* level-set values (65^3, flat (i*np+j)*np+k) of one subdomain given final densities; returns particle count of the subdomain or -1 if unoccupied */
int64_t so_debug_levelset_subdomain(const float *xyz, uint64_t n, const so_params *params,
                                    int64_t flat_subdomain, float *out_grid);
/* kernel.rs:58-107 */
float so_cubic_kernel_evaluate(float compact_support_radius, float r);
/* marching_cubes_lut.rs (emitted order), 256x16 */
const int8_t *so_mc_table(void);
/* dense_subdomains.rs:1810-1905: writes up to cap flat subdomain indices, returns count */
int so_classify_particle(const so_grid *subdomain_grid, float ghost_margin, const float p[3],
                         int64_t *out, int cap);

/* ---- sharded (multi-process) variant used by the world_size-2 gloo tests: restatement of what
 *      include/splashsurf_hip.h's ss_shard_* entry points compute, on the CPU ---- */
typedef struct so_shard {
    float domain_min[3]; /* AABB of all particles of the job */
    float domain_max[3];
    int64_t sub_lo[3];   /* half-open box of subdomain indices handled by this process */
    int64_t sub_hi[3];
} so_shard;
int so_grid_for_domain(const so_params *params, const float domain_min[3], const float domain_max[3], so_grid *grid,
                       so_grid *subdomain_grid, float *ghost_margin);
/* densities of the particles whose subdomain lies in the box (others 0) */
int so_shard_densities(const float *xyz, uint64_t n, const so_params *params, const so_shard *shard, float *rho_out);
/* level set + MC + stitching of the box's subdomains given densities of all local particles */
int so_shard_reconstruct(const float *xyz, uint64_t n, const so_params *params, const so_shard *shard, const float *rho,
                         so_result *out);

/* level-set values ((n+1)^3) of one subdomain of the shard for GIVEN densities: the stage-level
 * observable of density_grid_loop_scalar (dense_subdomains.rs:784-847); returns the subdomain's particle
 * count or -1 if it has none */
int64_t so_debug_shard_levelset(const float *xyz, uint64_t n, const so_params *params, const so_shard *shard, const float *rho,
                                int64_t flat_subdomain, float *out_grid);

#ifdef __cplusplus
}
#endif
#endif
