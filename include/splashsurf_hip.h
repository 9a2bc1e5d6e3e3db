/* splashsurf_hip.h -- C ABI of the MI355X-native surface reconstruction (libsplashsurf_hip.so).
 *
 * Drop-in boundary for the reference's Rust API (all citations relative to
 * /root/reference/splashsurf_lib/src/):
 *
 *   pub fn reconstruct_surface<I, R>(particle_positions: &[Vector3<R>], parameters: &Parameters<R>)
 *       -> Result<SurfaceReconstruction<I, R>, ReconstructionError<I, R>>          lib.rs:330-337
 *   pub fn reconstruct_surface_inplace(.., output_surface: &mut SurfaceReconstruction) lib.rs:340-473
 *   pub fn grid_for_reconstruction(..) -> Result<UniformGrid<I, R>, ..>             lib.rs:476-516
 *   pub fn initialize_thread_pool(num_threads)                                     lib.rs:321-326
 *
 * for the instantiation <I = i64, R = f32> (splashsurf/src/reconstruct.rs:982-1007,
 * pysplashsurf/src/utils.rs:9).  A Rust host binds these functions in an `extern "C"` block
 * (INTEGRATION.md shows the stub); this repository's own hosts are C++ (tests/bench harness) and
 * Python/ctypes (splashsurf_amd/api.py).
 *
 * Conventions: plain pointers and sizes only.  `xyz` is N x 3 contiguous f32 (the memory layout of
 * `&[Vector3<f32>]`) and may be a HOST or a DEVICE (HBM) pointer -- the library asks the HIP runtime.
 * Results are owned by the library (`ss_result`), live in HBM and are copied to (pinned) host
 * memory lazily by the accessors that return host pointers; free with `ss_result_free`.
 * One context may be used by one thread at a time; several contexts (also on several GPUs) may
 * coexist.  All functions return an `ss_status`; `ss_last_error` gives the message.
 *
 * Stream ordering: every context runs on its own non-blocking HIP stream (or the one given to
 * ss_context_set_stream) and synchronises it before returning, so OUTPUTS are complete on return.  DEVICE-pointer
 * INPUTS must be complete before the call: whatever produced them on another stream has to have finished (or the
 * library has to be put on that stream with ss_context_set_stream).  splashsurf_amd/api.py synchronises torch's
 * current stream before handing a tensor over.
 */
#ifndef SPLASHSURF_HIP_H
#define SPLASHSURF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 6

/* Return codes; 1..4 mirror ReconstructionError (lib.rs:289-314). */
typedef enum ss_status {
    SS_OK = 0,
    SS_ERR_GRID_CONSTRUCTION = 1,    /* GridConstructionError (uniform_grid.rs:147-169) */
    SS_ERR_DENSITY_MAP = 2,          /* DensityMapGenerationError */
    SS_ERR_MARCHING_CUBES = 3,       /* MarchingCubesError */
    SS_ERR_UNKNOWN = 4,              /* anyhow::Error / panics of the reference (e.g. cube_size <= 0, density_map.rs:555-559) */
    SS_ERR_DEVICE = 5,               /* HIP runtime / RCCL failure (no counterpart in the reference) */
    SS_ERR_INVALID_ARGUMENT = 6,
    SS_ERR_UNSUPPORTED = 7           /* feature of the reference not (yet) provided by this build */
} ss_status;

/* Sub-codes of SS_ERR_GRID_CONSTRUCTION, see ss_last_error_detail (uniform_grid.rs:147-169) */
enum {
    SS_GRID_INVALID_CELL_SIZE = 1,
    SS_GRID_DEGENERATE_AABB = 2,
    SS_GRID_INCONSISTENT_AABB = 3,
    SS_GRID_INDEX_TYPE_TOO_SMALL = 4
};

/* Parameters<f32> (lib.rs:158-189) + SpatialDecomposition/GridDecompositionParameters (lib.rs:121-154) */
typedef struct ss_params_f32 {
    float particle_radius;
    float rest_density;
    float compact_support_radius; /* absolute distance units */
    float cube_size;              /* absolute distance units */
    float iso_surface_threshold;
    int32_t has_particle_aabb;    /* Option<Aabb3d> */
    float aabb_min[3];
    float aabb_max[3];
    int32_t enable_multi_threading; /* accepted for API parity; the GPU path is always parallel.  Results of the global
                                     * strategy are those of the reference with enable_multi_threading = false (its only
                                     * deterministic mode, reconstruction.rs:65-194) */
    int32_t enable_simd;            /* Parameters::enable_simd (lib.rs:179-181; the reference's default is true), f32 only -- the
                                     * reference's SIMD loop exists for <i64, f32> alone (dense_subdomains.rs:1413-1415):
                                     * 0  level set by the scalar loop (dense_subdomains.rs:784-847, kernel.rs:58-107), bit-identical
                                     *    to the reference with enable_simd = false;
                                     * 1  the arithmetic of the AVX2+FMA loop (dense_subdomains.rs:991-1133, kernel.rs:319-378) for
                                     *    EVERY (particle, grid point) pair: d^2 by two fma, support d^2 < h^2, correctly rounded sqrt,
                                     *    q = r * (1/h), v = max(1 - q, 0), fma polynomial with sigma = 8/(pi h^3), fma accumulate.
                                     *    The reference itself mixes three arithmetics under this flag (vector lanes, unfused
                                     *    remainder lanes, the scalar loop for sparse subdomains), so its values on shared subdomain
                                     *    faces depend on the subdomain; this library computes every global grid point once.  Meshes
                                     *    agree with the reference's enable_simd = true output in topology and to ~1e-6 relative in
                                     *    vertex positions;
                                     * 2  (extension) as 1 with the hardware v_sqrt_f32 (<= 1 ulp) instead of the correctly rounded
                                     *    root: the fastest mode, same tolerance class.
                                     * Particle densities never depend on this flag (they do not in the reference either). */
    int32_t decomposition;          /* 0 = SpatialDecomposition::None (global strategy, reconstruction.rs:65-112),
                                     * 1 = UniformGrid (subdomain grid, the optimised path) */
    uint32_t subdomain_num_cubes_per_dim; /* default 64 */
    int32_t auto_disable;           /* GridDecompositionParameters::auto_disable: global strategy when the domain has
                                     * <= (1.2 n) cells per dimension (lib.rs:421-441) */
    int32_t global_neighborhood_list;
} ss_params_f32;

/* UniformGrid<i64, f32> (uniform_grid.rs:128-142) */
typedef struct ss_grid_f32 {
    float aabb_min[3];
    float aabb_max[3];
    float cell_size;
    int64_t n_points[3];
    int64_t n_cells[3];
} ss_grid_f32;

/* The same for the reference's f64 instantiation (reconstruct_surface::<i64, f64>; pysplashsurf dispatches
 * on the array dtype, pysplashsurf/src/reconstruction.rs:187-206; the CLI has --double-precision). */
typedef struct ss_params_f64 {
    double particle_radius;
    double rest_density;
    double compact_support_radius;
    double cube_size;
    double iso_surface_threshold;
    int32_t has_particle_aabb;
    double aabb_min[3];
    double aabb_max[3];
    int32_t enable_multi_threading;
    int32_t enable_simd;
    int32_t decomposition;
    uint32_t subdomain_num_cubes_per_dim;
    int32_t auto_disable;
    int32_t global_neighborhood_list;
} ss_params_f64;

typedef struct ss_grid_f64 {
    double aabb_min[3];
    double aabb_max[3];
    double cell_size;
    int64_t n_points[3];
    int64_t n_cells[3];
} ss_grid_f64;

/* Stage timings (milliseconds, HIP events on the context's stream) with the reference's profiling
 * scope names (README.md:198-231; dense_subdomains.rs `profile!` scopes) plus device-specific rows. */
typedef struct ss_stats {
    double ms_total;                  /* "surface reconstruction subdomain-grid" */
    double ms_upload;                 /* H2D of particle positions (0 if xyz was a device pointer) */
    double ms_aabb_grid;              /* "compute minimum enclosing aabb" + grid set-up */
    double ms_decomposition;          /* "decomposition": cell binning + stable sort */
    double ms_density;                /* "compute_global_density_vector" */
    double ms_levelset;               /* "reconstruction" / density grid loop (level-set splat kernel(s) only) */
    double ms_levelset_prepare;       /* active-block detection + compaction for the splat */
    double ms_marching_cubes;         /* "reconstruction" / mc triangulation loop */
    double ms_stitching;              /* "stitching": global vertex numbering (prefix sums) */
    uint64_t n_particles;             /* after the AABB filter */
    uint64_t n_vertices;
    uint64_t n_triangles;
    uint64_t n_active_blocks;         /* level-set blocks of 8^3 points evaluated */
    uint64_t n_block_candidates;      /* sum over blocks of candidate particles (tile sizes) */
    uint64_t fast_div_verified;       /* 1 if the splat used the exhaustively verified reciprocal division for this h */
    uint64_t levelset_kernel_launches;
    uint64_t bytes_device_peak;       /* HBM held by the context after this call */
    double ms_levelset_gather;        /* part of ms_levelset: the arena path of over-dense blocks (k_splat_bounds, offsets, k_splat_gather[_large]); 0 without such blocks */
    double ms_levelset_accumulate;    /* part of ms_levelset: k_splat_fused (+ k_splat_accumulate_list for over-dense blocks), both passes: the dominant kernel */
    uint64_t n_large_tile_blocks;     /* blocks whose candidate tile (> 384 entries) was ordered by the workgroup-level gather */
    uint64_t arith_mode;              /* arithmetic of the level-set accumulation that ran: 0 scalar (generic sqrt/divide), 1 scalar
                                       * (lean exact sqrt + verified reciprocal division), 2 / 3 SIMD with correctly rounded sqrt
                                       * (generic / lean), 4 SIMD with v_sqrt_f32 */
    uint64_t bytes_tile_arena;        /* bytes of the candidate tiles of all blocks (16 B x candidates within reach; in LDS, over-dense blocks in the arena) */
    uint64_t bytes_tile_arena_reserved; /* size of the arena of the over-dense blocks' tiles (ranges sized by a cheap per-block upper bound; 0 without such blocks) */
    uint64_t n_certified_subblocks;   /* 4x4x4 sub-blocks the classification pass of the splat certified to lie inside the fluid */
    uint64_t n_truncated_blocks;      /* active blocks left with truncated (lower-bound) level-set values: inside the fluid, never read by MC */
    uint64_t n_completed_blocks;      /* truncated blocks next to the surface that the second splat pass evaluated in full */
    double ms_levelset_accumulate_pass2; /* part of ms_levelset_accumulate: k_select_redo + the second launch of k_splat_fused */
    uint64_t n_mc_blocks;             /* blocks of 8^3 cells marching cubes visited (their 2x2x2 level-set blocks straddle the threshold) */
    double ms_density_kernel;         /* part of ms_density: k_density_sub (the neighbourhood search + SPH sums themselves) */
    double ms_mc_count;               /* part of ms_marching_cubes: k_mc_count (classification, crossing masks, counts) */
    double ms_mc_emit;                /* part of ms_marching_cubes: k_mc_emit (vertices, keys, triangles) */
    uint64_t n_host_waits;            /* points of the call at which the host waited for the device before it could enqueue more work: counts polled from
                                       * mail slots (waits on slots that are posted together count once), device-to-host copies of a count, the final drain */
} ss_stats;

typedef struct ss_context ss_context;
typedef struct ss_result ss_result;

/* -- context: replaces initialize_thread_pool + the reconstruction workspace (lib.rs:321-326, workspace.rs) -- */
int ss_abi_version(void);
ss_status ss_context_create(int device_id, ss_context **out);
void ss_context_destroy(ss_context *ctx);
const char *ss_last_error(const ss_context *ctx);
int ss_last_error_detail(const ss_context *ctx);
/* Context options.  SS_OPTION_FULL_LEVELSET (default 0): 1 = evaluate the level set completely at every grid point of every
 * active block.  By default the splat first certifies 4x4x4 sub-blocks that lie inside the fluid with a lower bound of the level
 * set (a sum over the nearby particles only; all terms are >= 0) and evaluates in full what is not certified plus the certified
 * sub-blocks with a point next to a grid point outside the surface -- the end points of the edges marching cubes interpolates on:
 * mesh, densities and every level-set value that influences them are unchanged; the values of the other certified sub-blocks are
 * never computed or stored (ss_result_levelset_box reports an error while such blocks exist).  Set it before
 * ss_result_levelset_box is used to inspect values away from the surface.
 * SS_OPTION_SPLAT_TWO_PASS (default -1): -1 = the library decides per workload whether the certification scheme above pays off
 * (jobs below 1 k active blocks and workloads whose previous call certified < 30 % of the sub-blocks evaluate everything), 0 = never,
 * 1 = always (tests).  Output is identical in every setting; with -1 the time of a call depends on what the previous calls on this
 * context certified (a workload is re-probed every 16th call).
 * SS_OPTION_WIDEN_ON_DEVICE (default 0): ss_result_triangles hands out [usize; 3] = u64 indices; for meshes of a million indices and more
 * they cross PCIe as u32 in chunks and host threads widen them into the pinned buffer while the next chunk is in flight; 1 = widen on
 * the device and copy 8 bytes per index (what small meshes always do).
 * SS_OPTION_SPLIT_MC_OFFSETS (default 0): the vertex and triangle offsets of the marching-cubes blocks come out of ONE prefix sum over
 * packed 31 + 31 bit counts while the worst case of the totals fits, and out of two 64-bit prefix sums for larger jobs (more than
 * 838 860 surface blocks); 1 = always the two-sum form (tests).  Output is identical. */
enum { SS_OPTION_FULL_LEVELSET = 1, SS_OPTION_SPLAT_TWO_PASS = 2, SS_OPTION_WIDEN_ON_DEVICE = 3, SS_OPTION_SPLIT_MC_OFFSETS = 4 };
ss_status ss_context_set_option(ss_context *ctx, int option, int value);
/* use an existing HIP stream (hipStream_t passed as void*); NULL = context's own stream */
ss_status ss_context_set_stream(ss_context *ctx, void *hip_stream);
/* Measurement aid: the HBM rate this device sustains for a float4 read stream and for a float4 copy (read + write bytes counted) over
 * buffers of `bytes` each (use >= 1 GiB: beyond the 256 MiB Infinity Cache), best of `repetitions` -- the measured peak beside which
 * bench.py quotes the splat kernel's roofline fraction; GB/s = 1e9 bytes per second. */
ss_status ss_measure_hbm_bandwidth(ss_context *ctx, uint64_t bytes, int repetitions, double *read_gbs, double *copy_gbs);

/* -- the boundary -- */
ss_status ss_reconstruct_surface_f32(ss_context *ctx, const float *xyz, uint64_t n_particles,
                                     const ss_params_f32 *params, ss_result **out);
/* reuses the buffers held by `inout` (obtained from ss_result_create or a previous call) */
ss_status ss_reconstruct_surface_inplace_f32(ss_context *ctx, const float *xyz, uint64_t n_particles,
                                             const ss_params_f32 *params, ss_result *inout);
ss_status ss_grid_for_reconstruction_f32(ss_context *ctx, const float *xyz, uint64_t n_particles,
                                         const ss_params_f32 *params, ss_grid_f32 *out);

/* f64 instantiation of the boundary (xyz: N x 3 contiguous f64, host or device) */
ss_status ss_reconstruct_surface_f64(ss_context *ctx, const double *xyz, uint64_t n_particles,
                                     const ss_params_f64 *params, ss_result **out);
ss_status ss_reconstruct_surface_inplace_f64(ss_context *ctx, const double *xyz, uint64_t n_particles,
                                             const ss_params_f64 *params, ss_result *inout);
ss_status ss_grid_for_reconstruction_f64(ss_context *ctx, const double *xyz, uint64_t n_particles,
                                         const ss_params_f64 *params, ss_grid_f64 *out);

ss_status ss_result_create(ss_context *ctx, ss_result **out);
void ss_result_free(ss_result *res);

/* -- SurfaceReconstruction accessors (lib.rs:247-262); host pointers stay valid until the result is
 *    reused or freed -- */
ss_status ss_result_counts(const ss_result *res, uint64_t *n_vertices, uint64_t *n_triangles);
ss_status ss_result_vertices(ss_result *res, const float **xyz, uint64_t *n_vertices);          /* mesh.vertices */
ss_status ss_result_triangles(ss_result *res, const uint64_t **indices, uint64_t *n_triangles); /* mesh.triangles as [usize;3] */
ss_status ss_result_triangles_u32(ss_result *res, const uint32_t **indices, uint64_t *n_triangles);
ss_status ss_result_grid(const ss_result *res, ss_grid_f32 *out);
/* *present = 1 if the subdomain grid was used, 0 (Option::None) after the global strategy (lib.rs:249-250) */
ss_status ss_result_subdomain_grid(const ss_result *res, ss_grid_f32 *out, int32_t *present);
ss_status ss_result_particle_densities(ss_result *res, const float **rho, uint64_t *n);
/* *flags == NULL when no particle AABB was given (Option::None) */
ss_status ss_result_particle_inside_aabb(ss_result *res, const uint8_t **flags, uint64_t *n);
/* SurfaceReconstruction::particle_neighbors (lib.rs:256-257) as CSR: neighbours of particle i are
 * neighbors[row_ptr[i] .. row_ptr[i+1]), global particle indices in the reference's order
 * (dense_subdomains.rs:617-639).  *row_ptr == NULL when global_neighborhood_list was not requested. */
ss_status ss_result_particle_neighbors(ss_result *res, const uint64_t **row_ptr, const uint64_t **neighbors, uint64_t *n_particles);
ss_status ss_result_stats(const ss_result *res, ss_stats *out);
/* 1 if the result holds an f64 reconstruction (then only the *_f64 value accessors apply), else 0 */
int ss_result_is_f64(const ss_result *res);
/* value accessors of an f64 result (the index/flag/stat accessors above are type independent) */
ss_status ss_result_vertices_f64(ss_result *res, const double **xyz, uint64_t *n_vertices);
ss_status ss_result_particle_densities_f64(ss_result *res, const double **rho, uint64_t *n);
ss_status ss_result_grid_f64(const ss_result *res, ss_grid_f64 *out);
ss_status ss_result_subdomain_grid_f64(const ss_result *res, ss_grid_f64 *out, int32_t *present);
ss_status ss_result_levelset_box_f64(ss_result *res, const int64_t lo[3], const int64_t extent[3], double *out);

/* -- device-side views (HBM pointers; no copy) -- */
ss_status ss_result_device_vertices(const ss_result *res, const float **d_xyz, uint64_t *n_vertices);
ss_status ss_result_device_triangles_u32(const ss_result *res, const uint32_t **d_indices, uint64_t *n_triangles);
ss_status ss_result_device_particle_densities(const ss_result *res, const float **d_rho, uint64_t *n);

/* -- extensions used by the parity tests (no counterpart in the reference API) -- */
/* per-vertex global grid edge key = ((gi*NPy+gj)*NPz+gk)*3 + axis, NP = grid.n_points */
ss_status ss_result_vertex_keys(ss_result *res, const uint64_t **keys, uint64_t *n_vertices);
/* Level-set values of the box of grid points [lo, lo+extent) of the global MC grid of the last
 * reconstruction held by `res` (points outside evaluated blocks read 0); out is host memory,
 * extent[0]*extent[1]*extent[2] floats, k fastest (dense_subdomains.rs:839 flattening). */
ss_status ss_result_levelset_box(ss_result *res, const int64_t lo[3], const int64_t extent[3], float *out);
/* Test aid: the certificates of the last subdomain-grid reconstruction held by `res` (valid until the next call on its context: SS_ERR_INVALID_ARGUMENT
 * after that, the masks live in the context's scratch).  n_active: number of
 * active 8^3 level-set blocks; masks[b] bit s set <=> the 4^3 sub-block s = (sx << 2 | sy << 1 | sz) of block b was certified "inside the fluid" by the
 * lower bound and NEVER evaluated (SS_OPTION_FULL_LEVELSET above); block_xyz[3 b ..]: the block's coordinates in units of blocks.  `capacity`: blocks the
 * two arrays hold (masks: capacity, block_xyz: 3 * capacity; host memory); n_active is always reported.  tests/test_gpu_certificates.py checks every such
 * sub-block against the completely evaluated level set: all 64 values must lie above the threshold. */
ss_status ss_result_debug_certified(ss_result *res, uint32_t *masks, uint32_t *block_xyz, uint64_t capacity, uint64_t *n_active);
/* The reference's decomposition statistics for the last reconstruction (number of occupied
 * subdomains S and sum of per-subdomain particle counts incl. ghosts, dense_subdomains.rs:349-494);
 * only needed to price the splat kernel in the reference's algorithmic bytes. */
ss_status ss_result_subdomain_stats(ss_result *res, uint64_t *n_occupied_subdomains, uint64_t *n_subdomain_particles);

/* -- frame pipeline: a time series of frames through `depth` contexts of ONE device (csrc/ss_pipeline.hip) --
 * The reference's time-series use is a loop of reconstruct_surface_inplace over the frames with one workspace (lib.rs:340-346, 466-470).  A
 * host-to-host frame of this library is a chain (upload -> kernels -> mesh download -> index widening) in which nothing of one frame overlaps;
 * the pipeline overlaps CONSECUTIVE frames instead: every slot owns a context (stream, workspace), a result and a host thread, frame t runs
 * on slot t % depth, and while frame t's mesh crosses PCIe frame t + 1 uploads and computes.  Output of a frame is what
 * ss_reconstruct_surface_inplace_f32 on a context of its own returns (tests/test_gpu_pipeline.py: bit-identical).  The canonical loop:
 *     submit(f0); submit(f1); for (k = 0; k < n; ++k) { next(&res); consume(res); if (k + 2 < n) submit(f[k+2]); }
 *   ss_pipeline_submit_*   queues a frame and returns at once.  xyz (host or device memory, N x 3) must stay valid and unchanged until the frame
 *                          was handed back by ss_pipeline_next; the parameters are copied.  `fetch`: SS_FETCH_* bits of the host mirrors the
 *                          slot's thread fills before the frame counts as complete (the matching ss_result_* accessors then return without
 *                          copying).  SS_ERR_INVALID_ARGUMENT while `depth` frames are in flight.
 *   ss_pipeline_next       blocks until the OLDEST frame in flight is complete and hands it back: its status (message: ss_pipeline_last_error),
 *                          its ticket and -- on success -- its result.  The result belongs to the pipeline and is valid until the submit that
 *                          reuses its slot (the `depth`-th submit after the frame's own); never pass it to ss_result_free.
 *   ss_pipeline_ready      1 if ss_pipeline_next would return without blocking.
 *   ss_pipeline_set_option ss_context_set_option on every slot's context; only while no frame is in flight.
 *   ss_pipeline_context    the context of a slot (e.g. for ss_last_error after a failing accessor of that slot's result); NULL if out of range.
 *   ss_pipeline_frame_times  host wall time of the last frame of a slot: the reconstruct call and the fetches behind it.
 * One thread drives a pipeline (submit / next / destroy are not re-entrant); the frames themselves run on the slots' threads. */
typedef struct ss_pipeline ss_pipeline;
#define SS_PIPELINE_MAX_DEPTH 8
enum { SS_FETCH_VERTICES = 1, SS_FETCH_TRIANGLES_U64 = 2, SS_FETCH_TRIANGLES_U32 = 4, SS_FETCH_DENSITIES = 8 };
ss_status ss_pipeline_create(int device_id, int depth, ss_pipeline **out);
void ss_pipeline_destroy(ss_pipeline *p); /* waits for the frames in flight */
const char *ss_pipeline_last_error(const ss_pipeline *p);
int ss_pipeline_depth(const ss_pipeline *p);
int ss_pipeline_in_flight(const ss_pipeline *p);
ss_context *ss_pipeline_context(ss_pipeline *p, int slot);
ss_status ss_pipeline_set_option(ss_pipeline *p, int option, int value);
ss_status ss_pipeline_submit_f32(ss_pipeline *p, const float *xyz, uint64_t n_particles, const ss_params_f32 *params, uint32_t fetch, uint64_t *ticket);
ss_status ss_pipeline_submit_f64(ss_pipeline *p, const double *xyz, uint64_t n_particles, const ss_params_f64 *params, uint32_t fetch, uint64_t *ticket);
ss_status ss_pipeline_next(ss_pipeline *p, ss_result **result, uint64_t *ticket);
int ss_pipeline_ready(ss_pipeline *p);
ss_status ss_pipeline_frame_times(ss_pipeline *p, int slot, double *ms_reconstruct, double *ms_fetch);

/* -- multi-GPU extension (no counterpart in the single-process reference) --
 * One process per GPU reconstructs a box of subdomains of ONE global grid.  The host (e.g.
 * splashsurf_amd/distributed.py over torch.distributed/RCCL) hands each process every particle within
 * the ghost margin of its box, in ascending GLOBAL particle order:
 *   ss_shard_begin_f32   grid of the whole job from `domain_min/max` (AABB of all particles), binning,
 *                        densities of the particles contained in the box's subdomains (others: 0);
 *   [host exchanges densities: ss_shard_get_densities -> all-reduce -> ss_shard_set_densities]
 *   ss_shard_finish      level set, marching cubes and numbering for the box; vertices on the box's
 *                        faces are emitted by both neighbours with identical keys and coordinates.
 * No other call may be made on the context between the two phases. */
typedef struct ss_shard_f32 {
    float domain_min[3]; /* AABB of ALL particles of the job, as Aabb3d::par_from_points would give (aabb.rs:28-52) */
    float domain_max[3];
    int64_t sub_lo[3];   /* half-open box of subdomain indices reconstructed by this process */
    int64_t sub_hi[3];
} ss_shard_f32;
typedef struct ss_shard_f64 {
    double domain_min[3];
    double domain_max[3];
    int64_t sub_lo[3];
    int64_t sub_hi[3];
} ss_shard_f64;
ss_status ss_shard_begin_f32(ss_context *ctx, const float *xyz, uint64_t n_particles, const ss_params_f32 *params,
                             const ss_shard_f32 *shard, ss_result *inout);
ss_status ss_shard_begin_f64(ss_context *ctx, const double *xyz, uint64_t n_particles, const ss_params_f64 *params,
                             const ss_shard_f64 *shard, ss_result *inout);
ss_status ss_shard_finish(ss_context *ctx, ss_result *inout);
/* densities of the particles passed to ss_shard_begin_*, in their order; dst/src may be host or device memory */
ss_status ss_shard_get_densities(ss_result *res, float *dst, uint64_t n);
ss_status ss_shard_set_densities(ss_result *res, const float *src, uint64_t n);
ss_status ss_shard_get_densities_f64(ss_result *res, double *dst, uint64_t n);
ss_status ss_shard_set_densities_f64(ss_result *res, const double *src, uint64_t n);
/* host-only helper: global MC grid, subdomain grid and ghost margin for a given particle AABB
 * (lib.rs:476-516 + dense_subdomains.rs:89-244); lets every rank derive the same partition */
ss_status ss_grid_for_domain_f32(const ss_params_f32 *params, const float domain_min[3], const float domain_max[3],
                                 ss_grid_f32 *grid, ss_grid_f32 *subdomain_grid, float *ghost_margin);
ss_status ss_grid_for_domain_f64(const ss_params_f64 *params, const double domain_min[3], const double domain_max[3],
                                 ss_grid_f64 *grid, ss_grid_f64 *subdomain_grid, double *ghost_margin);

/* -- multi-GPU, native (csrc/ss_dist.hip): the whole sharded reconstruction of SURVEY.md section 8e behind the C ABI --
 * One process (or host thread) per GPU calls ss_dist_reconstruct_* with ITS share of the particles (any split; global
 * particle order = concatenation by rank); the library cuts the subdomain grid (dense_subdomains.rs:349-494, the unit the
 * reference parallelises over, :1582-1598) into `world` bricks balanced by particle count, moves halo positions and halo
 * densities between the ranks itself -- grouped ncclSend/ncclRecv and small ncclAllGather/ncclAllReduce on the context's
 * stream, RCCL over xGMI -- and runs ss_shard_begin / ss_shard_finish on the brick.  ss_dist_assemble then numbers the mesh
 * globally: a vertex on a brick face belongs to the lowest rank holding its edge (the rule of globalize_local_edge,
 * dense_subdomains.rs:1260-1329), the other holders learn its global id from the owner (the join of `stitching`,
 * dense_subdomains.rs:1693-1733), triangles carry global ids; the mesh is the concatenation over ranks of (owned vertices,
 * triangles).  Densities, level set and mesh are bit-identical to a single-GPU call on the union of the inputs.
 * A communicator is bound to one context; collective calls must be made by every rank.  A peer that never shows up makes the
 * call fail with SS_ERR_DEVICE after SPLASH_COMM_TIMEOUT_S seconds (default 120) instead of hanging. */
typedef struct ss_comm ss_comm;
#define SS_COMM_ID_BYTES 128
/* rank 0 creates an id (ncclGetUniqueId) and hands it to the other ranks out of band (MPI, a socket, a file, torch's store) */
ss_status ss_comm_unique_id(uint8_t id[SS_COMM_ID_BYTES]);
ss_status ss_comm_create_rccl(ss_context *ctx, const uint8_t id[SS_COMM_ID_BYTES], int rank, int world, ss_comm **out);
/* adopt a communicator the host already owns (ncclComm_t passed as void*); ss_comm_destroy leaves it alive */
ss_status ss_comm_adopt_rccl(ss_context *ctx, void *nccl_comm, int rank, int world, ss_comm **out);
/* in-process group without RCCL: `world` communicators for `world` contexts on ONE device, each driven by its own host
 * thread (how the tests run the whole algorithm on a single-GPU box; RCCL refuses two ranks on one device) */
ss_status ss_comm_create_local_group(ss_context *const *ctxs, int world, ss_comm **out /* world entries */);
/* Measurement aid for in-process groups (call it on any member before the threads start): with on = 1 a rank holds the shared device
 * exclusively while it computes between two exchange steps and drains its stream before handing it on, so that the per-rank stage
 * timers (ss_result_stats, ss_dist_info) read what the rank takes on a GPU of its own instead of the time it spent queueing behind the
 * other ranks' kernels.  Results are unaffected. */
ss_status ss_comm_local_group_take_turns(ss_comm *comm, int on);
/* Partition feedback for time series (default off): with on = 1 every ss_dist_reconstruct after the first weighs the owner histogram of a subdomain with the
 * cost per owned particle (ms_device / n_owned) measured in the previous call by the rank that owned it, so that the bricks balance measured cost instead of
 * particle counts (surface-heavy bricks cost more per particle).  Collective: every rank of the communicator sets the same value.  The mesh does not depend
 * on the partition. */
ss_status ss_comm_set_balance_feedback(ss_comm *comm, int on);
void ss_comm_destroy(ss_comm *comm);

typedef struct ss_dist_info {
    int32_t rank, world;
    int64_t brick_lo[3], brick_hi[3];    /* this rank's half-open box of subdomain indices */
    uint64_t n_total;                    /* particles of the whole job */
    uint64_t n_held;                     /* particles this rank holds (owned + ghosts) */
    uint64_t n_owned;                    /* particles contained in this rank's brick */
    uint64_t bytes_sent_positions, bytes_sent_densities, bytes_sent_assembly; /* payload this rank sent to OTHER ranks */
    double ms_partition, ms_position_exchange, ms_density_exchange, ms_assembly; /* host wall time incl. device waits and waits for peers */
    double ms_phase1, ms_phase2;         /* host time of the two phases of the rank's own reconstruction (binning + densities; level set + marching cubes).  ms_phase1 is the
                                            time to ENQUEUE phase 1 plus its in-phase count waits: its kernels are not drained at the end of the phase, so their tail is part of
                                            ms_density_exchange (the first wait that follows); ms_device has the phases' device time from HIP events */
    double ms_own_turns;                 /* ss_comm_local_group_take_turns only: time this rank held the device (all of its own work of the step, exchanges excluded) */
    uint64_t n_vertices_owned, vertex_offset, n_vertices_total;  /* after ss_dist_assemble */
    uint64_t n_triangles, triangle_offset, n_triangles_total;
    uint64_t n_collectives;              /* communication steps of the last ss_dist_reconstruct + ss_dist_assemble in which this rank met its peers (all-gathers, the all-reduce,
                                            grouped send/recv): each costs a collective's latency on a real multi-GPU node, which the one-GPU projection of bench.py prices */
    double ms_device;                    /* HIP-event time of this rank's two reconstruction phases (ss_result_stats ms_total): the cost the partition feedback balances */
    uint64_t bytes_link_max;             /* sum over the step's three exchanges of the largest payload this rank exchanged with ONE peer in one direction (max over peers of
                                            max(sent, received)): on a fully connected xGMI node every pair of GPUs has its own link, so this, not the total, sets an exchange's transfer time */
} ss_dist_info;

/* xyz_local: this rank's n_local x 3 particles (host or HBM).  On return `inout` holds the brick's reconstruction: its mesh with
 * local indices (ss_result_*), densities of the held particles (ss_result_particle_densities: owned + received halo values). */
ss_status ss_dist_reconstruct_f32(ss_comm *comm, const float *xyz_local, uint64_t n_local, const ss_params_f32 *params, ss_result *inout);
ss_status ss_dist_reconstruct_f64(ss_comm *comm, const double *xyz_local, uint64_t n_local, const ss_params_f64 *params, ss_result *inout);
ss_status ss_dist_assemble(ss_comm *comm, ss_result *inout);
ss_status ss_dist_get_info(const ss_comm *comm, ss_dist_info *out);
/* host-only: the partition rule by itself -- recursive bisection of the subdomain grid (owner-particle count per subdomain,
 * x-major like the subdomain grid) into `world` bricks, bricks[6*world] = (lo[3], hi[3]); axis_preference breaks ties
 * between equally good cuts (smaller = preferred; NULL: none) */
ss_status ss_dist_partition(const uint32_t *histogram, const int64_t n_subdomains[3], int world, const double axis_preference[3], int64_t *bricks);
/* the partition of the last call, identical on every rank: bricks[6*world] = (lo[3], hi[3]) per rank, owned / held particle
 * counts per rank (any pointer may be NULL) */
ss_status ss_dist_get_partition(const ss_comm *comm, int64_t *bricks, uint64_t *owned, uint64_t *held);
/* copies into caller buffers (host or HBM): global ids of the held particles (ascending, n_held); after ss_dist_assemble the
 * owned vertices (n_vertices_owned x 3, the job's Real type), their global edge keys, and this rank's triangles with GLOBAL
 * vertex ids (n_triangles x 3 uint64) */
ss_status ss_dist_copy_global_ids(ss_comm *comm, uint64_t *dst);
ss_status ss_dist_copy_vertices(ss_comm *comm, void *dst);
ss_status ss_dist_copy_vertex_keys(ss_comm *comm, uint64_t *dst);
ss_status ss_dist_copy_triangles(ss_comm *comm, uint64_t *dst);

/* -- stand-alone entry points on the stages of the global strategy --
 * marching_cubes::triangulate_density_map on a dense array of function values (marching_cubes.rs:100-127;
 * pysplashsurf.marching_cubes): values[(i*ny + j)*nz + k] at grid point translation + (i,j,k)*cube_size, host or HBM
 * pointer; the mesh is read through the ss_result accessors (vertices by ascending edge key). */
ss_status ss_marching_cubes_f32(ss_context *ctx, const float *values, const int64_t n_points[3], float iso_surface_threshold, float cube_size,
                                const float translation[3] /* NULL: origin */, ss_result *inout);
ss_status ss_marching_cubes_f64(ss_context *ctx, const double *values, const int64_t n_points[3], double iso_surface_threshold, double cube_size,
                                const double translation[3], ss_result *inout);
/* neighborhood_search::neighborhood_search_spatial_hashing (neighborhood_search.rs:131-230; pysplashsurf exposes the
 * parallel variant, whose per-cell order depends on thread timing): lists via ss_result_particle_neighbors, in the
 * order of the reference's sequential function.  Particles outside `domain` are an error (the reference panics). */
ss_status ss_neighborhood_search_f32(ss_context *ctx, const float *xyz, uint64_t n_particles, const float domain_min[3], const float domain_max[3],
                                     float search_radius, ss_result *inout);
ss_status ss_neighborhood_search_f64(ss_context *ctx, const double *xyz, uint64_t n_particles, const double domain_min[3], const double domain_max[3],
                                     double search_radius, ss_result *inout);

/* =====================================================================================================
 * Post-processing: the stages of the reference's pipeline that consume the mesh right after the
 * reconstruction (splashsurf/src/reconstruct.rs:1085-1345).  Every array argument may be a host pointer or an HBM
 * pointer (e.g. ss_result_device_vertices); HBM pointers are used in place.  Triangles are uint32 triples (the
 * device-native index type, ss_result_triangles_u32 / ss_result_device_triangles_u32).
 * ===================================================================================================== */

/* TriMesh3d::vertex_vertex_connectivity (splashsurf_lib/src/mesh.rs:290-306) as CSR: the neighbours of vertex i are
 * neighbors[row_ptr[i] .. row_ptr[i+1]) in the reference's first-occurrence order.  row_ptr: n_vertices + 1 entries;
 * neighbors: caller-allocated, capacity 6 * n_triangles always suffices; *n_entries = entries written (or required). */
ss_status ss_post_vertex_connectivity(ss_context *ctx, uint64_t n_vertices, const uint32_t *triangles, uint64_t n_triangles, uint64_t *row_ptr,
                                      uint32_t *neighbors, uint64_t neighbors_capacity, uint64_t *n_entries);
/* TriMesh3d::vertex_normals (mesh.rs:782-796, 868-886): area-weighted, normalised; summation in triangle order */
ss_status ss_post_vertex_normals_f32(ss_context *ctx, const float *vertices, uint64_t n_vertices, const uint32_t *triangles, uint64_t n_triangles, float *normals);
ss_status ss_post_vertex_normals_f64(ss_context *ctx, const double *vertices, uint64_t n_vertices, const uint32_t *triangles, uint64_t n_triangles, double *normals);
/* postprocessing::par_laplacian_smoothing_inplace (splashsurf_lib/src/postprocessing.rs:17-52); weights == NULL: all 1 */
ss_status ss_post_laplacian_smoothing_f32(ss_context *ctx, float *vertices, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *neighbors, uint32_t iterations,
                                          float beta, const float *weights);
ss_status ss_post_laplacian_smoothing_f64(ss_context *ctx, double *vertices, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *neighbors, uint32_t iterations,
                                          double beta, const double *weights);
/* postprocessing::par_laplacian_smoothing_normals_inplace (postprocessing.rs:55-96) */
ss_status ss_post_smooth_normals_f32(ss_context *ctx, float *normals, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *neighbors, uint32_t iterations);
ss_status ss_post_smooth_normals_f64(ss_context *ctx, double *normals, uint64_t n_vertices, const uint64_t *row_ptr, const uint32_t *neighbors, uint32_t iterations);
/* distance-weighted neighbour count per particle (splashsurf/src/reconstruct.rs:1189-1204); neighbour lists as CSR with
 * uint32 indices (ss_result_device_particle_neighbors) */
ss_status ss_post_weighted_neighbor_counts_f32(ss_context *ctx, const float *xyz, uint64_t n, const uint64_t *nb_row_ptr, const uint32_t *nb_indices, float h, float *out);
ss_status ss_post_weighted_neighbor_counts_f64(ss_context *ctx, const double *xyz, uint64_t n, const uint64_t *nb_row_ptr, const uint32_t *nb_indices, double h, double *out);
/* smoothing weights from interpolated counts: clamp, / normalization, smooth-step (reconstruct.rs:1219-1232) */
ss_status ss_post_smoothing_weights_f32(ss_context *ctx, const float *wnn, uint64_t n, float normalization, float *out);
ss_status ss_post_smoothing_weights_f64(ss_context *ctx, const double *wnn, uint64_t n, double normalization, double *out);
/* SphInterpolator::interpolate_{scalar,vector}_quantity (splashsurf_lib/src/sph_interpolation.rs:205-259); dim = 1 or 3.
 * Sums run over a uniform cell grid (the reference: an rstar R-tree), i.e. same terms, other order: ~1e-6 relative in f32 */
ss_status ss_post_sph_interpolate_f32(ss_context *ctx, const float *xyz, const float *rho, uint64_t n, float rest_mass, float h, const float *values, int32_t dim,
                                      const float *points, uint64_t n_points, int32_t first_order_correction, float *out);
ss_status ss_post_sph_interpolate_f64(ss_context *ctx, const double *xyz, const double *rho, uint64_t n, double rest_mass, double h, const double *values, int32_t dim,
                                      const double *points, uint64_t n_points, int32_t first_order_correction, double *out);
/* SphInterpolator::interpolate_normals (sph_interpolation.rs:72-113) */
ss_status ss_post_sph_normals_f32(ss_context *ctx, const float *xyz, const float *rho, uint64_t n, float rest_mass, float h, const float *points, uint64_t n_points,
                                  float *out);
ss_status ss_post_sph_normals_f64(ss_context *ctx, const double *xyz, const double *rho, uint64_t n, double rest_mass, double h, const double *points, uint64_t n_points,
                                  double *out);
/* HBM views of the neighbour lists of a reconstruction (NULL when absent): CSR rows (uint64) and uint32 indices */
ss_status ss_result_device_particle_neighbors(const ss_result *res, const uint64_t **row_ptr, const uint32_t **neighbors, uint64_t *n_particles, uint64_t *n_entries);
/* copies of the reconstruction's arrays into caller buffers (host or HBM; element type = the result's Real type):
 * the post-processing stages modify their arrays in place while the ss_result keeps the raw mesh */
ss_status ss_result_copy_vertices(ss_result *res, void *dst);                  /* n_vertices x 3 */
ss_status ss_result_copy_triangles_u32(ss_result *res, uint32_t *dst);         /* n_triangles x 3 */
ss_status ss_result_copy_particle_densities(ss_result *res, void *dst);        /* n_particles */
ss_status ss_result_copy_vertex_keys(ss_result *res, uint64_t *dst);           /* n_vertices global edge keys (ss_result_vertex_keys) */

#ifdef __cplusplus
}
#endif
#endif /* SPLASHSURF_HIP_H */
