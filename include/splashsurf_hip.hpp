// splashsurf_hip.hpp -- header-only C++17 host over the C ABI (splashsurf_hip.h).
//
// Mirrors the reference's Rust API for this path so that C++ callers (and the parity tests in
// tests/cpp/) read like the reference's own code (citations: /root/reference/splashsurf_lib/src/):
//
//   splashsurf::Parameters                      <->  Parameters<f32>            lib.rs:158-243
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

using Vector3f = std::array<float, 3>;

struct Aabb3d {
    Vector3f min{0, 0, 0}, max{0, 0, 0};
};

struct GridDecompositionParameters {  // lib.rs:139-154
    uint32_t subdomain_num_cubes_per_dim = 64;
    bool auto_disable = true;
};

struct SpatialDecomposition {  // lib.rs:121-136 (default: UniformGrid)
    enum class Kind { None, UniformGrid } kind = Kind::UniformGrid;
    GridDecompositionParameters grid{};
};

struct Parameters {  // lib.rs:158-189
    float particle_radius = 0.0f;
    float rest_density = 1000.0f;
    float compact_support_radius = 0.0f;
    float cube_size = 0.0f;
    float iso_surface_threshold = 0.6f;
    std::optional<Aabb3d> particle_aabb;
    bool enable_multi_threading = true;
    bool enable_simd = true;
    SpatialDecomposition spatial_decomposition{};
    bool global_neighborhood_list = false;

    // Parameters::new (lib.rs:197-210): absolute units
    static Parameters with(float particle_radius, float compact_support_radius, float cube_size) {
        Parameters p;
        p.particle_radius = particle_radius;
        p.compact_support_radius = compact_support_radius;
        p.cube_size = cube_size;
        return p;
    }
    // Parameters::new_relative (lib.rs:216-226)
    static Parameters relative(float particle_radius, float relative_compact_support_radius, float relative_cube_size) {
        return with(particle_radius, particle_radius * relative_compact_support_radius, particle_radius * relative_cube_size);
    }

    ss_params_f32 to_c() const {
        ss_params_f32 c{};
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

struct UniformGrid {  // uniform_grid.rs:128-142
    Aabb3d aabb;
    float cell_size = 0.0f;
    std::array<int64_t, 3> points_per_dim{0, 0, 0}, cells_per_dim{0, 0, 0};
    static UniformGrid from_c(const ss_grid_f32& g) {
        UniformGrid u;
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

struct TriMesh3d {  // mesh.rs:187-193
    std::vector<Vector3f> vertices;
    std::vector<std::array<uint64_t, 3>> triangles;  // [usize; 3]
};

class ReconstructionError : public std::runtime_error {  // lib.rs:289-314
  public:
    enum class Variant { GridConstruction = 1, DensityMapGeneration = 2, MarchingCubes = 3, Unknown = 4, Device = 5, InvalidArgument = 6, Unsupported = 7 };
    ReconstructionError(int status, int detail, const std::string& msg)
        : std::runtime_error(msg), variant(static_cast<Variant>(status)), detail(detail) {}
    Variant variant;
    int detail;  // GridConstructionError sub-code (uniform_grid.rs:147-169)
};

struct SurfaceReconstruction {  // lib.rs:247-262
    UniformGrid grid;
    std::optional<UniformGrid> subdomain_grid;
    std::optional<std::vector<float>> particle_densities;
    std::optional<std::vector<bool>> particle_inside_aabb;
    std::optional<std::vector<std::vector<uint64_t>>> particle_neighbors;
    TriMesh3d mesh;
    ss_stats stats{};
};

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

    // reconstruct_surface::<i64, f32> (lib.rs:330-337)
    SurfaceReconstruction reconstruct_surface(const std::vector<Vector3f>& particle_positions, const Parameters& parameters) {
        SurfaceReconstruction out;
        reconstruct_surface_inplace(particle_positions, parameters, out);
        return out;
    }

    // reconstruct_surface_inplace (lib.rs:340-473): clears and refills `output_surface`, device/pinned buffers are reused
    void reconstruct_surface_inplace(const std::vector<Vector3f>& particle_positions, const Parameters& parameters,
                                     SurfaceReconstruction& output_surface) {
        const ss_params_f32 p = parameters.to_c();
        const float* xyz = particle_positions.empty() ? nullptr : particle_positions[0].data();
        check(ss_reconstruct_surface_inplace_f32(ctx_, xyz, particle_positions.size(), &p, res_));
        ss_grid_f32 g{};
        check(ss_result_grid(res_, &g));
        output_surface.grid = UniformGrid::from_c(g);
        int32_t present = 0;
        check(ss_result_subdomain_grid(res_, &g, &present));
        if (present)
            output_surface.subdomain_grid = UniformGrid::from_c(g);
        else
            output_surface.subdomain_grid.reset();
        const float* v = nullptr;
        uint64_t nv = 0;
        check(ss_result_vertices(res_, &v, &nv));
        output_surface.mesh.vertices.resize(nv);
        for (uint64_t i = 0; i < nv; ++i) output_surface.mesh.vertices[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
        const uint64_t* t = nullptr;
        uint64_t nt = 0;
        check(ss_result_triangles(res_, &t, &nt));
        output_surface.mesh.triangles.resize(nt);
        for (uint64_t i = 0; i < nt; ++i) output_surface.mesh.triangles[i] = {t[3 * i], t[3 * i + 1], t[3 * i + 2]};
        const float* rho = nullptr;
        uint64_t n = 0;
        check(ss_result_particle_densities(res_, &rho, &n));
        output_surface.particle_densities = std::vector<float>(rho, rho + n);
        const uint8_t* inside = nullptr;
        uint64_t ni = 0;
        check(ss_result_particle_inside_aabb(res_, &inside, &ni));
        if (inside)
            output_surface.particle_inside_aabb = std::vector<bool>(inside, inside + ni);
        else
            output_surface.particle_inside_aabb.reset();
        const uint64_t *row = nullptr, *nb = nullptr;
        uint64_t np = 0;
        check(ss_result_particle_neighbors(res_, &row, &nb, &np));
        if (row) {
            std::vector<std::vector<uint64_t>> lists(np);
            for (uint64_t i = 0; i < np; ++i) lists[i].assign(nb + row[i], nb + row[i + 1]);
            output_surface.particle_neighbors = std::move(lists);
        } else {
            output_surface.particle_neighbors.reset();
        }
        check(ss_result_stats(res_, &output_surface.stats));
    }

    // grid_for_reconstruction (lib.rs:476-516)
    UniformGrid grid_for_reconstruction(const std::vector<Vector3f>& particle_positions, const Parameters& parameters) {
        const ss_params_f32 p = parameters.to_c();
        ss_grid_f32 g{};
        const float* xyz = particle_positions.empty() ? nullptr : particle_positions[0].data();
        check(ss_grid_for_reconstruction_f32(ctx_, xyz, particle_positions.size(), &p, &g));
        return UniformGrid::from_c(g);
    }

    ss_context* raw() { return ctx_; }

  private:
    void check(ss_status st) {
        if (st != SS_OK) throw ReconstructionError(st, ss_last_error_detail(ctx_), ss_last_error(ctx_));
    }
    ss_context* ctx_ = nullptr;
    ss_result* res_ = nullptr;
};

}  // namespace splashsurf
