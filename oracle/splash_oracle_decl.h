/* splash_oracle_decl.h -- declarations of the CPU oracle for one Real type; included twice by
 * splash_oracle.h (SO_REAL = float with prefix so_, SO_REAL = double with prefix so64_). */
typedef struct SOT(params) {
    SO_REAL particle_radius;
    SO_REAL rest_density;
    SO_REAL compact_support_radius; /* absolute units (lib.rs:163-164) */
    SO_REAL cube_size;              /* absolute units (lib.rs:165-166) */
    SO_REAL iso_surface_threshold;
    int32_t has_particle_aabb;
    SO_REAL aabb_min[3];
    SO_REAL aabb_max[3];
    int32_t subdomain_num_cubes_per_dim; /* lib.rs:142, default 64 */
    int32_t num_threads;                 /* <=0: all cores (OpenMP) */
    int32_t global_neighborhood_list;    /* lib.rs:185-188 */
    int32_t global_strategy; /* 0: subdomain grid (UniformGrid, auto_disable=false); 1: SpatialDecomposition::None;
                                2: UniformGrid with auto_disable=true (lib.rs:419-462) */
    int32_t enable_simd;     /* Parameters::enable_simd (lib.rs:179-181).  0: every subdomain runs density_grid_loop_scalar.
                                1: f32 only, DENSE subdomains run the restatement of density_grid_loop_avx
                                (dense_subdomains.rs:991-1133) incl. its unfused remainder lanes; sparse subdomains
                                (<= max(100, max_particles/20) particles, :1248-1253, :1590-1596) stay scalar.
                                2 (no counterpart in the reference): the AVX arithmetic applied uniformly to EVERY
                                (particle, grid point) pair -- all lanes fused, no sparse/dense distinction -- which is
                                what one value per global grid point means; correctly rounded sqrt. */
} SOT(params);

typedef struct SOT(grid) {
    SO_REAL aabb_min[3];
    SO_REAL aabb_max[3];
    SO_REAL cell_size;
    int64_t n_points[3];
    int64_t n_cells[3];
} SOT(grid);

typedef struct SOT(result) {
    SOT(grid) grid;            /* padded global MC grid (reconstruction.rs:27-29) */
    SOT(grid) subdomain_grid;  /* reconstruction.rs:26 */
    uint64_t n_input;        /* input particle count */
    uint64_t n_particles;    /* after the optional AABB filter */
    SO_REAL *particle_densities;      /* [n_particles] */
    uint8_t *particle_inside_aabb;  /* [n_input] or NULL when no AABB given */
    uint64_t *neighbor_ptr;         /* [n_particles+1] CSR rows or NULL (dense_subdomains.rs:617-639) */
    uint64_t *neighbors;            /* global particle indices */
    uint64_t n_vertices;
    SO_REAL *vertices;         /* [n_vertices*3] */
    uint64_t *vertex_keys;   /* [n_vertices] global edge key = ((gi*NPy+gj)*NPz+gk)*3+axis */
    uint64_t n_triangles;
    uint64_t *triangles;     /* [n_triangles*3] */
    int64_t n_subdomains;    /* occupied subdomains (patches) */
    uint64_t n_subdomain_particles; /* sum over subdomains incl. ghosts */
    double t_total, t_decomposition, t_density, t_reconstruction, t_stitching; /* seconds */
    int32_t threads_used;
    int32_t used_global_strategy;  /* 1: reconstruct_surface_global ran (subdomain_grid is all zero = None) */
    SO_REAL *global_levelset;      /* global strategy only: dense level-set values over grid.n_points (missing map entries = 0) */
} SOT(result);

/* returns 0 on success; 1 grid construction error; 4 other */
int SOFN(reconstruct_surface)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params, SOT(result) *out);
void SOFN(result_free)(SOT(result) *r);

/* lib.rs:476-516 */
int SOFN(grid_for_reconstruction)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params, SOT(grid) *out);

/* stand-alone marching cubes on a dense value array (marching_cubes.rs:100-127); 0 ok, 1 grid error, 3 triangulation error */
int SOFN(marching_cubes)(const SO_REAL *values, const int64_t n_points[3], SO_REAL threshold, SO_REAL cube_size, const SO_REAL translation[3],
                         SOT(result) *out);

/* Debug/observability entry points used by the parity tests */
/* level-set values (65^3, flat (i*np+j)*np+k) of one subdomain given final densities; returns particle count of the subdomain or -1 if unoccupied */
int64_t SOFN(debug_levelset_subdomain)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params,
                                    int64_t flat_subdomain, SO_REAL *out_grid);
/* kernel.rs:58-107 */
SO_REAL SOFN(cubic_kernel_evaluate)(SO_REAL compact_support_radius, SO_REAL r);
/* marching_cubes_lut.rs (emitted order), 256x16 */
const int8_t *SOFN(mc_table)(void);
/* dense_subdomains.rs:1810-1905: writes up to cap flat subdomain indices, returns count */
int SOFN(classify_particle)(const SOT(grid) *subdomain_grid, SO_REAL ghost_margin, const SO_REAL p[3],
                         int64_t *out, int cap);

/* ---- sharded (multi-process) variant used by the world_size-2 gloo tests: restatement of what
 *      include/splashsurf_hip.h's ss_shard_* entry points compute, on the CPU ---- */
typedef struct SOT(shard) {
    SO_REAL domain_min[3]; /* AABB of all particles of the job */
    SO_REAL domain_max[3];
    int64_t sub_lo[3];   /* half-open box of subdomain indices handled by this process */
    int64_t sub_hi[3];
} SOT(shard);
int SOFN(grid_for_domain)(const SOT(params) *params, const SO_REAL domain_min[3], const SO_REAL domain_max[3], SOT(grid) *grid,
                       SOT(grid) *subdomain_grid, SO_REAL *ghost_margin);
/* densities of the particles whose subdomain lies in the box (others 0) */
int SOFN(shard_densities)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params, const SOT(shard) *shard, SO_REAL *rho_out);
/* level set + MC + stitching of the box's subdomains given densities of all local particles */
int SOFN(shard_reconstruct)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params, const SOT(shard) *shard, const SO_REAL *rho,
                         SOT(result) *out);

/* level-set values ((n+1)^3) of one subdomain of the shard for GIVEN densities: the stage-level
 * observable of density_grid_loop_scalar (dense_subdomains.rs:784-847); returns the subdomain's particle
 * count or -1 if it has none */
int64_t SOFN(debug_shard_levelset)(const SO_REAL *xyz, uint64_t n, const SOT(params) *params, const SOT(shard) *shard, const SO_REAL *rho,
                                int64_t flat_subdomain, SO_REAL *out_grid);

