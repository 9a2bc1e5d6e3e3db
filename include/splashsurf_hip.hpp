// splashsurf_hip.hpp -- header-only C++17 host over the C ABI (splashsurf_hip.h).
//
// Mirrors the reference's Rust API for this path so that C++ callers (and the parity tests in
// tests/cpp/) read like the reference's own code (citations: /root/reference/splashsurf_lib/src/):
//
// Every type is a template on the Real type (float / double) with the f32 instantiation under the plain name:
//   splashsurf::Parameters = ParametersT<float>  <->  Parameters<f32>;  ParametersT<double> <-> Parameters<f64>   lib.rs:158-243
//   splashsurf::SpatialDecomposition            <->  SpatialDecomposition       lib.rs:121-154
//   splashsurf::UniformGrid / Aabb3d            <->  UniformGrid<i64,f32>       uniform_grid.rs:128-142
//   splashsurf::TriMesh3d                       <->  TriMesh3d<f32>             mesh.rs:187-193
//   splashsurf::SurfaceReconstruction           <->  SurfaceReconstruction      lib.rs:247-262
//   splashsurf::ReconstructionError (exception) <->  ReconstructionError        lib.rs:289-314
//   splashsurf::Context::reconstruct_surface    <->  reconstruct_surface        lib.rs:330-337
//   splashsurf::Context::reconstruct_surface_inplace <-> reconstruct_surface_inplace lib.rs:340-473
//   splashsurf::Context::grid_for_reconstruction <-> grid_for_reconstruction    lib.rs:476-516
//
// Rust's `Result<_, ReconstructionError>` becomes a C++ exception carrying the same variant.
#pragma once

#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "splashsurf_hip.h"

namespace splashsurf {

// C-ABI bindings per Real type (reconstruct_surface::<i64, f32> / ::<i64, f64>, reconstruct.rs:982-1007)
template <class R> struct Abi;
template <> struct Abi<float> {
    using params = ss_params_f32;
    using grid = ss_grid_f32;
    static ss_status reconstruct_inplace(ss_context* c, const float* xyz, uint64_t n, const params* p, ss_result* r) { return ss_reconstruct_surface_inplace_f32(c, xyz, n, p, r); }
    static ss_status grid_for_reconstruction(ss_context* c, const float* xyz, uint64_t n, const params* p, grid* g) { return ss_grid_for_reconstruction_f32(c, xyz, n, p, g); }
    static ss_status result_grid(const ss_result* r, grid* g) { return ss_result_grid(r, g); }
    static ss_status result_subdomain_grid(const ss_result* r, grid* g, int32_t* present) { return ss_result_subdomain_grid(r, g, present); }
    static ss_status vertices(ss_result* r, const float** v, uint64_t* n) { return ss_result_vertices(r, v, n); }
    static ss_status densities(ss_result* r, const float** v, uint64_t* n) { return ss_result_particle_densities(r, v, n); }
};
template <> struct Abi<double> {
    using params = ss_params_f64;
    using grid = ss_grid_f64;
    static ss_status reconstruct_inplace(ss_context* c, const double* xyz, uint64_t n, const params* p, ss_result* r) { return ss_reconstruct_surface_inplace_f64(c, xyz, n, p, r); }
    static ss_status grid_for_reconstruction(ss_context* c, const double* xyz, uint64_t n, const params* p, grid* g) { return ss_grid_for_reconstruction_f64(c, xyz, n, p, g); }
    static ss_status result_grid(const ss_result* r, grid* g) { return ss_result_grid_f64(r, g); }
    static ss_status result_subdomain_grid(const ss_result* r, grid* g, int32_t* present) { return ss_result_subdomain_grid_f64(r, g, present); }
    static ss_status vertices(ss_result* r, const double** v, uint64_t* n) { return ss_result_vertices_f64(r, v, n); }
    static ss_status densities(ss_result* r, const double** v, uint64_t* n) { return ss_result_particle_densities_f64(r, v, n); }
};

template <class R> using Vector3 = std::array<R, 3>;
using Vector3f = Vector3<float>;
using Vector3d = Vector3<double>;

template <class R>
struct Aabb3dT {
    Vector3<R> min{0, 0, 0}, max{0, 0, 0};
};
using Aabb3d = Aabb3dT<float>;

struct GridDecompositionParameters {  // lib.rs:139-154
    uint32_t subdomain_num_cubes_per_dim = 64;
    bool auto_disable = true;
};

struct SpatialDecomposition {  // lib.rs:121-136 (default: UniformGrid)
    enum class Kind { None, UniformGrid } kind = Kind::UniformGrid;
    GridDecompositionParameters grid{};
};

template <class R>
struct ParametersT {  // lib.rs:158-189
    R particle_radius = R(0.0);
    R rest_density = R(1000.0);
    R compact_support_radius = R(0.0);
    R cube_size = R(0.0);
    R iso_surface_threshold = R(0.6);
    std::optional<Aabb3dT<R>> particle_aabb;
    bool enable_multi_threading = true;
    bool enable_simd = true;
    SpatialDecomposition spatial_decomposition{};
    bool global_neighborhood_list = false;

    // Parameters::new (lib.rs:197-210): absolute units
    static ParametersT with(R particle_radius, R compact_support_radius, R cube_size) {
        ParametersT p;
        p.particle_radius = particle_radius;
        p.compact_support_radius = compact_support_radius;
        p.cube_size = cube_size;
        return p;
    }
    // Parameters::new_relative (lib.rs:216-226)
    static ParametersT relative(R particle_radius, R relative_compact_support_radius, R relative_cube_size) {
        return with(particle_radius, particle_radius * relative_compact_support_radius, particle_radius * relative_cube_size);
    }

    typename Abi<R>::params to_c() const {
        typename Abi<R>::params c{};
        c.particle_radius = particle_radius;
        c.rest_density = rest_density;
        c.compact_support_radius = compact_support_radius;
        c.cube_size = cube_size;
        c.iso_surface_threshold = iso_surface_threshold;
        c.has_particle_aabb = particle_aabb ? 1 : 0;
        if (particle_aabb)
            for (int d = 0; d < 3; ++d) {
                c.aabb_min[d] = particle_aabb->min[d];
                c.aabb_max[d] = particle_aabb->max[d];
            }
        c.enable_multi_threading = enable_multi_threading;
        c.enable_simd = enable_simd;
        c.decomposition = spatial_decomposition.kind == SpatialDecomposition::Kind::UniformGrid ? 1 : 0;
        c.subdomain_num_cubes_per_dim = spatial_decomposition.grid.subdomain_num_cubes_per_dim;
        c.auto_disable = spatial_decomposition.grid.auto_disable;
        c.global_neighborhood_list = global_neighborhood_list;
        return c;
    }
};
using Parameters = ParametersT<float>;

template <class R>
struct UniformGridT {  // uniform_grid.rs:128-142
    Aabb3dT<R> aabb;
    R cell_size = R(0.0);
    std::array<int64_t, 3> points_per_dim{0, 0, 0}, cells_per_dim{0, 0, 0};
    static UniformGridT from_c(const typename Abi<R>::grid& g) {
        UniformGridT u;
        for (int d = 0; d < 3; ++d) {
            u.aabb.min[d] = g.aabb_min[d];
            u.aabb.max[d] = g.aabb_max[d];
            u.points_per_dim[d] = g.n_points[d];
            u.cells_per_dim[d] = g.n_cells[d];
        }
        u.cell_size = g.cell_size;
        return u;
    }
};
using UniformGrid = UniformGridT<float>;

template <class R>
struct TriMesh3dT {  // mesh.rs:187-193
    std::vector<Vector3<R>> vertices;
    std::vector<std::array<uint64_t, 3>> triangles;  // [usize; 3]
};
using TriMesh3d = TriMesh3dT<float>;

class ReconstructionError : public std::runtime_error {  // lib.rs:289-314
  public:
    enum class Variant { GridConstruction = 1, DensityMapGeneration = 2, MarchingCubes = 3, Unknown = 4, Device = 5, InvalidArgument = 6, Unsupported = 7 };
    ReconstructionError(int status, int detail, const std::string& msg)
        : std::runtime_error(msg), variant(static_cast<Variant>(status)), detail(detail) {}
    Variant variant;
    int detail;  // GridConstructionError sub-code (uniform_grid.rs:147-169)
};

template <class R>
struct SurfaceReconstructionT {  // lib.rs:247-262
    UniformGridT<R> grid;
    std::optional<UniformGridT<R>> subdomain_grid;
    std::optional<std::vector<R>> particle_densities;
    std::optional<std::vector<bool>> particle_inside_aabb;
    std::optional<std::vector<std::vector<uint64_t>>> particle_neighbors;
    TriMesh3dT<R> mesh;
    ss_stats stats{};
};
using SurfaceReconstruction = SurfaceReconstructionT<float>;

// fills a SurfaceReconstruction (lib.rs:247-262) from the accessors of a completed result; `check_` turns a status into an exception
template <class R, class Check>
void unpack_result(ss_result* res, Check check_, SurfaceReconstructionT<R>& output_surface) {
    typename Abi<R>::grid g{};
    check_(Abi<R>::result_grid(res, &g));
    output_surface.grid = UniformGridT<R>::from_c(g);
    int32_t present = 0;
    check_(Abi<R>::result_subdomain_grid(res, &g, &present));
    if (present)
        output_surface.subdomain_grid = UniformGridT<R>::from_c(g);
    else
        output_surface.subdomain_grid.reset();
    const R* v = nullptr;
    uint64_t nv = 0;
    check_(Abi<R>::vertices(res, &v, &nv));
    output_surface.mesh.vertices.resize(nv);
    for (uint64_t i = 0; i < nv; ++i) output_surface.mesh.vertices[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    const uint64_t* t = nullptr;
    uint64_t nt = 0;
    check_(ss_result_triangles(res, &t, &nt));
    output_surface.mesh.triangles.resize(nt);
    for (uint64_t i = 0; i < nt; ++i) output_surface.mesh.triangles[i] = {t[3 * i], t[3 * i + 1], t[3 * i + 2]};
    const R* rho = nullptr;
    uint64_t n = 0;
    check_(Abi<R>::densities(res, &rho, &n));
    output_surface.particle_densities = std::vector<R>(rho, rho + n);
    const uint8_t* inside = nullptr;
    uint64_t ni = 0;
    check_(ss_result_particle_inside_aabb(res, &inside, &ni));
    if (inside)
        output_surface.particle_inside_aabb = std::vector<bool>(inside, inside + ni);
    else
        output_surface.particle_inside_aabb.reset();
    const uint64_t *row = nullptr, *nb = nullptr;
    uint64_t np = 0;
    check_(ss_result_particle_neighbors(res, &row, &nb, &np));
    if (row) {
        std::vector<std::vector<uint64_t>> lists(np);
        for (uint64_t i = 0; i < np; ++i) lists[i].assign(nb + row[i], nb + row[i + 1]);
        output_surface.particle_neighbors = std::move(lists);
    } else {
        output_surface.particle_neighbors.reset();
    }
    check_(ss_result_stats(res, &output_surface.stats));
}

class Context {  // replaces initialize_thread_pool (lib.rs:321-326) + the reconstruction workspace
  public:
    explicit Context(int device_id = 0) {
        ss_status st = ss_context_create(device_id, &ctx_);
        if (st != SS_OK) throw ReconstructionError(st, 0, "ss_context_create failed: no usable HIP device");
        st = ss_result_create(ctx_, &res_);
        if (st != SS_OK) {
            ss_context_destroy(ctx_);
            throw ReconstructionError(st, 0, "ss_result_create failed");
        }
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ~Context() {
        if (res_) ss_result_free(res_);
        if (ctx_) ss_context_destroy(ctx_);
    }

    // reconstruct_surface::<i64, R> (lib.rs:330-337), R = float or double
    template <class R>
    SurfaceReconstructionT<R> reconstruct_surface(const std::vector<Vector3<R>>& particle_positions, const ParametersT<R>& parameters) {
        SurfaceReconstructionT<R> out;
        reconstruct_surface_inplace(particle_positions, parameters, out);
        return out;
    }

    // reconstruct_surface_inplace (lib.rs:340-473): clears and refills `output_surface`, device/pinned buffers are reused
    template <class R>
    void reconstruct_surface_inplace(const std::vector<Vector3<R>>& particle_positions, const ParametersT<R>& parameters,
                                     SurfaceReconstructionT<R>& output_surface) {
        const typename Abi<R>::params p = parameters.to_c();
        const R* xyz = particle_positions.empty() ? nullptr : particle_positions[0].data();
        check(Abi<R>::reconstruct_inplace(ctx_, xyz, particle_positions.size(), &p, res_));
        unpack_result<R>(res_, [this](ss_status st) { check(st); }, output_surface);
    }

    // grid_for_reconstruction (lib.rs:476-516)
    template <class R>
    UniformGridT<R> grid_for_reconstruction(const std::vector<Vector3<R>>& particle_positions, const ParametersT<R>& parameters) {
        const typename Abi<R>::params p = parameters.to_c();
        typename Abi<R>::grid g{};
        const R* xyz = particle_positions.empty() ? nullptr : particle_positions[0].data();
        check(Abi<R>::grid_for_reconstruction(ctx_, xyz, particle_positions.size(), &p, &g));
        return UniformGridT<R>::from_c(g);
    }

    // f32 conveniences so that braced initialiser lists keep working: ctx.reconstruct_surface({{0, 0, 0}}, p)
    SurfaceReconstruction reconstruct_surface(const std::vector<Vector3f>& particle_positions, const Parameters& parameters) {
        return reconstruct_surface<float>(particle_positions, parameters);
    }
    UniformGrid grid_for_reconstruction(const std::vector<Vector3f>& particle_positions, const Parameters& parameters) {
        return grid_for_reconstruction<float>(particle_positions, parameters);
    }

    ss_context* raw() { return ctx_; }

  private:
    void check(ss_status st) {
        if (st != SS_OK) throw ReconstructionError(st, ss_last_error_detail(ctx_), ss_last_error(ctx_));
    }
    ss_context* ctx_ = nullptr;
    ss_result* res_ = nullptr;
};

// ---- time series: the reference's frame loop with `depth` frames in flight -----------------------------------------------
// `for frame in series { reconstruct_surface_inplace(&particles, &parameters, &mut output) }` (lib.rs:340-346: one workspace reused by every
// frame) as submit / next over ss_pipeline_*: every slot of the pipeline owns a context, a result and a host thread, so the upload and
// kernels of one frame run beside the mesh download of the frame before it.  Frames come back in submission order, each exactly what
// Context::reconstruct_surface_inplace gives.  The particle vector handed to submit() must stay alive and unchanged until next() returned
// that frame.
class FrameSeries {
  public:
    explicit FrameSeries(int device_id = 0, int depth = 2) {
        const ss_status st = ss_pipeline_create(device_id, depth, &pipe_);
        if (st != SS_OK) throw ReconstructionError(st, 0, "ss_pipeline_create failed: no usable HIP device, or depth outside 1..8");
    }
    FrameSeries(const FrameSeries&) = delete;
    FrameSeries& operator=(const FrameSeries&) = delete;
    ~FrameSeries() {
        if (pipe_) ss_pipeline_destroy(pipe_);
    }

    int depth() const { return ss_pipeline_depth(pipe_); }
    int in_flight() const { return ss_pipeline_in_flight(pipe_); }

    // queues a frame (returns at once) and gives its ticket; throws while depth() frames are in flight
    template <class R>
    uint64_t submit(const std::vector<Vector3<R>>& particle_positions, const ParametersT<R>& parameters) {
        const typename Abi<R>::params p = parameters.to_c();
        const R* xyz = particle_positions.empty() ? nullptr : particle_positions[0].data();
        uint64_t ticket = 0;
        const uint32_t fetch = SS_FETCH_VERTICES | SS_FETCH_TRIANGLES_U64 | SS_FETCH_DENSITIES;
        if constexpr (sizeof(R) == 4)
            check(ss_pipeline_submit_f32(pipe_, xyz, particle_positions.size(), &p, fetch, &ticket));
        else
            check(ss_pipeline_submit_f64(pipe_, xyz, particle_positions.size(), &p, fetch, &ticket));
        return ticket;
    }

    // blocks for the oldest frame in flight and fills `output_surface` with it (the Real type must be the one the frame was submitted with);
    // a failed frame throws its ReconstructionError here, the series goes on with the next frame
    template <class R>
    uint64_t next(SurfaceReconstructionT<R>& output_surface) {
        ss_result* res = nullptr;
        uint64_t ticket = 0;
        check(ss_pipeline_next(pipe_, &res, &ticket));
        ss_context* ctx = ss_pipeline_context(pipe_, (int)(ticket % (uint64_t)depth()));
        unpack_result<R>(res, [ctx](ss_status st) {
            if (st != SS_OK) throw ReconstructionError(st, ss_last_error_detail(ctx), ss_last_error(ctx));
        }, output_surface);
        return ticket;
    }

  private:
    void check(ss_status st) {
        if (st != SS_OK) throw ReconstructionError(st, 0, ss_pipeline_last_error(pipe_));
    }
    ss_pipeline* pipe_ = nullptr;
};

// ---- multi-GPU: one ShardedReconstruction per process (or host thread) and GPU ------------------------------------------
// The reference parallelises over subdomains inside one process (dense_subdomains.rs:1582-1598); across GPUs the library
// cuts the subdomain grid into bricks and exchanges halo particles, halo densities and the ids of shared vertices itself
// (ss_dist_*).  Every rank passes ITS share of the particles; the global particle order is the
// concatenation by rank.  Collective: every rank calls step() with the same parameters.
class ShardedReconstruction {
  public:
    using UniqueId = std::array<uint8_t, SS_COMM_ID_BYTES>;
    // rank 0 creates the id and hands it to the other ranks out of band (MPI_Bcast, a file, a socket)
    static UniqueId unique_id() {
        UniqueId id{};
        ss_status st = ss_comm_unique_id(id.data());
        if (st != SS_OK) throw ReconstructionError(st, 0, "ss_comm_unique_id failed: RCCL could not be loaded");
        return id;
    }
    ShardedReconstruction(int device_id, const UniqueId& id, int rank, int world) : ctx_(device_id) {
        ss_status st = ss_result_create(ctx_.raw(), &res_);
        if (st == SS_OK) st = ss_comm_create_rccl(ctx_.raw(), id.data(), rank, world, &comm_);
        if (st != SS_OK) {
            const ReconstructionError err(st, ss_last_error_detail(ctx_.raw()), ss_last_error(ctx_.raw()));
            if (res_) ss_result_free(res_);
            throw err;
        }
    }
    ShardedReconstruction(const ShardedReconstruction&) = delete;
    ShardedReconstruction& operator=(const ShardedReconstruction&) = delete;
    ~ShardedReconstruction() {
        if (comm_) ss_comm_destroy(comm_);
        if (res_) ss_result_free(res_);
    }

    // Time series: balance the bricks by the cost every rank measured in the previous frame instead of particle counts (with hysteresis; every
    // rank of the job sets the same value; the mesh does not depend on the partition).
    void set_balance_feedback(bool on) { check(ss_comm_set_balance_feedback(comm_, on ? 1 : 0)); }

    // Reconstruction of this rank's brick plus the global numbering of the mesh.  Returns the job's bookkeeping: brick, particle
    // counts, this rank's vertex / triangle offsets in the global mesh, bytes sent.
    template <class R>
    ss_dist_info step(const std::vector<Vector3<R>>& local_particles, const ParametersT<R>& parameters) {
        const typename Abi<R>::params p = parameters.to_c();
        const R* xyz = local_particles.empty() ? nullptr : local_particles[0].data();
        if constexpr (sizeof(R) == 4)
            check(ss_dist_reconstruct_f32(comm_, xyz, local_particles.size(), &p, res_));
        else
            check(ss_dist_reconstruct_f64(comm_, xyz, local_particles.size(), &p, res_));
        check(ss_dist_assemble(comm_, res_));
        ss_dist_info info{};
        check(ss_dist_get_info(comm_, &info));
        return info;
    }

    // This rank's piece of the global mesh: the vertices it owns (global ids vertex_offset .. + n_vertices_owned) and its triangles
    // with GLOBAL vertex ids; the mesh is the concatenation of the pieces over the ranks.
    template <class R>
    TriMesh3dT<R> mesh_piece() {
        ss_dist_info info{};
        check(ss_dist_get_info(comm_, &info));
        TriMesh3dT<R> m;
        m.vertices.resize(info.n_vertices_owned);
        m.triangles.resize(info.n_triangles);
        if (info.n_vertices_owned) check(ss_dist_copy_vertices(comm_, m.vertices[0].data()));
        if (info.n_triangles) check(ss_dist_copy_triangles(comm_, m.triangles[0].data()));
        return m;
    }

    // densities of the particles this rank holds (owned + ghosts) and their global ids (ascending)
    template <class R>
    void held_particles(std::vector<uint64_t>& global_ids, std::vector<R>& densities) {
        ss_dist_info info{};
        check(ss_dist_get_info(comm_, &info));
        global_ids.resize(info.n_held);
        if (info.n_held) check(ss_dist_copy_global_ids(comm_, global_ids.data()));
        const R* rho = nullptr;
        uint64_t n = 0;
        check(Abi<R>::densities(res_, &rho, &n));
        densities.assign(rho, rho + n);
    }

  private:
    void check(ss_status st) {
        if (st != SS_OK) throw ReconstructionError(st, ss_last_error_detail(ctx_.raw()), ss_last_error(ctx_.raw()));
    }
    Context ctx_;
    ss_result* res_ = nullptr;
    ss_comm* comm_ = nullptr;
};

}  // namespace splashsurf
