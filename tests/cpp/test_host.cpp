// C++ host test over the C ABI through include/splashsurf_hip.hpp.
// Mirrors the reference's integration test tests/integration_tests/test_simple.rs:71-126 (known-answer
// test: one particle => 6 vertices / 8 triangles, closed + manifold) and the error behaviour of
// lib.rs:289-314 / density_map.rs:555-559.  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <map>
#include <utility>

#include "splashsurf_hip.hpp"

using namespace splashsurf;

static int failures = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

// every directed edge once, its reverse once (what check_mesh_consistency asserts, marching_cubes.rs:129-213)
static bool closed_manifold(const TriMesh3d& m) {
    std::map<std::pair<uint64_t, uint64_t>, int> edges;
    for (const auto& t : m.triangles)
        for (int k = 0; k < 3; ++k) edges[{t[k], t[(k + 1) % 3]}]++;
    for (const auto& e : edges) {
        if (e.second != 1) return false;
        auto r = edges.find({e.first.second, e.first.first});
        if (r == edges.end() || r->second != 1) return false;
    }
    return true;
}

int main() {
    Context ctx(0);

    // --- test_simple.rs:71-126 ---
    {
        std::vector<Vector3f> particles = {{0.01f, 0.0f, 0.0f}};
        Parameters p = Parameters::with(1.0f, 1.0f, 1.0f);
        p.iso_surface_threshold = 0.1f;
        p.spatial_decomposition.grid.auto_disable = false;
        SurfaceReconstruction s = ctx.reconstruct_surface(particles, p);
        CHECK(s.mesh.vertices.size() == 6);
        CHECK(s.mesh.triangles.size() == 8);
        CHECK(closed_manifold(s.mesh));
        CHECK(s.particle_densities && s.particle_densities->size() == 1);
        CHECK(std::fabs((*s.particle_densities)[0] - 20371.834f) < 1e-2f);
        CHECK(s.subdomain_grid.has_value());
        CHECK(s.grid.cells_per_dim[0] == 64 && s.grid.aabb.min[0] == -2.0f && s.grid.aabb.min[1] == -3.0f);
        CHECK(!s.particle_inside_aabb.has_value());
        CHECK(!s.particle_neighbors.has_value());
        for (const auto& v : s.mesh.vertices) {
            const float r = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            CHECK(std::fabs(r - 0.89994f) < 2e-3f);
        }
        // in-place reuse (lib.rs:340-346) + neighbour lists
        std::vector<Vector3f> two = {{0.0f, 0.0f, 0.0f}, {0.5f, 0.0f, 0.0f}, {5.0f, 5.0f, 5.0f}};
        p.global_neighborhood_list = true;
        ctx.reconstruct_surface_inplace(two, p, s);
        CHECK(s.particle_neighbors.has_value() && s.particle_neighbors->size() == 3);
        CHECK((*s.particle_neighbors)[0].size() == 1 && (*s.particle_neighbors)[0][0] == 1);
        CHECK((*s.particle_neighbors)[1].size() == 1 && (*s.particle_neighbors)[1][0] == 0);
        CHECK((*s.particle_neighbors)[2].empty());
        CHECK(closed_manifold(s.mesh));
    }
    // --- the same known answer with the global strategy (test_simple.rs:71-126 runs both), then the auto-disable rule ---
    {
        std::vector<Vector3f> particles = {{0.01f, 0.0f, 0.0f}};
        Parameters p = Parameters::with(1.0f, 1.0f, 1.0f);
        p.iso_surface_threshold = 0.1f;
        p.spatial_decomposition.kind = SpatialDecomposition::Kind::None;
        SurfaceReconstruction s = ctx.reconstruct_surface(particles, p);
        CHECK(s.mesh.vertices.size() == 6);
        CHECK(s.mesh.triangles.size() == 8);
        CHECK(closed_manifold(s.mesh));
        CHECK(!s.subdomain_grid.has_value());                    // lib.rs:249-250
        CHECK(s.grid.cells_per_dim[0] == 5 && s.grid.cells_per_dim[1] == 6 && s.grid.aabb.min[0] == -2.0f);  // not padded to subdomains
        CHECK(s.particle_neighbors.has_value() && s.particle_neighbors->size() == 1);  // always Some for this strategy (reconstruction.rs:107-108)
        CHECK(std::fabs((*s.particle_densities)[0] - 20371.834f) < 1e-2f);
        p.spatial_decomposition.kind = SpatialDecomposition::Kind::UniformGrid;  // default auto_disable = true: 6 cells <= 76 -> global
        SurfaceReconstruction a = ctx.reconstruct_surface(particles, p);
        CHECK(!a.subdomain_grid.has_value() && a.mesh.vertices.size() == 6);
    }
    // --- reconstruct_surface::<i64, f64>: the known answer in double precision ---
    {
        std::vector<Vector3d> particles = {{0.01, 0.0, 0.0}};
        ParametersT<double> p = ParametersT<double>::with(1.0, 1.0, 1.0);
        p.iso_surface_threshold = 0.1;
        p.spatial_decomposition.grid.auto_disable = false;
        SurfaceReconstructionT<double> s = ctx.reconstruct_surface(particles, p);
        CHECK(s.mesh.vertices.size() == 6 && s.mesh.triangles.size() == 8);
        CHECK(s.particle_densities && std::fabs((*s.particle_densities)[0] - 20371.83271) < 1e-4);
        CHECK(s.grid.cells_per_dim[0] == 64 && s.grid.aabb.min[0] == -2.0);
        UniformGridT<double> g = ctx.grid_for_reconstruction(particles, p);
        CHECK(g.cells_per_dim[0] == 5);
    }
    // --- the sharded path over a one-rank RCCL communicator: same mesh as the single-context call ---
    {
        std::vector<Vector3f> particles;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j)
                for (int k = 0; k < 6; ++k) particles.push_back({0.05f * i, 0.05f * j, 0.05f * k});
        Parameters p = Parameters::relative(0.025f, 2.0f, 0.75f);
        p.spatial_decomposition.grid.auto_disable = false;
        p.spatial_decomposition.grid.subdomain_num_cubes_per_dim = 16;
        SurfaceReconstruction direct = ctx.reconstruct_surface(particles, p);
        ShardedReconstruction sharded(0, ShardedReconstruction::unique_id(), 0, 1);
        const ss_dist_info info = sharded.step<float>(particles, p);
        CHECK(info.world == 1 && info.n_total == particles.size() && info.n_owned == particles.size());
        CHECK(info.n_vertices_total == direct.mesh.vertices.size() && info.n_triangles_total == direct.mesh.triangles.size());
        TriMesh3d piece = sharded.mesh_piece<float>();
        CHECK(piece.vertices.size() == direct.mesh.vertices.size() && piece.triangles.size() == direct.mesh.triangles.size());
        CHECK(closed_manifold(piece));
        std::vector<uint64_t> ids;
        std::vector<float> rho;
        sharded.held_particles<float>(ids, rho);
        CHECK(ids.size() == particles.size() && rho.size() == particles.size() && ids.front() == 0 && ids.back() == particles.size() - 1);
        for (size_t i = 0; i < rho.size(); ++i) CHECK(rho[i] == (*direct.particle_densities)[i]);
    }
    // --- a time series through FrameSeries (ss_pipeline_*): the frame loop of lib.rs:340-346 with two frames in flight; every frame equals
    //     the one-context call, frames come back in order, a failing frame throws at next() and the series goes on ---
    {
        std::vector<std::vector<Vector3f>> frames(5);
        for (int f = 0; f < 5; ++f)
            for (int i = 0; i < 5 + f; ++i)
                for (int j = 0; j < 6; ++j)
                    for (int k = 0; k < 6; ++k) frames[f].push_back({0.05f * i + 0.004f * f, 0.05f * j, 0.05f * k - 0.002f * f});
        Parameters p = Parameters::relative(0.025f, 2.0f, 0.75f);
        p.spatial_decomposition.grid.auto_disable = false;
        p.spatial_decomposition.grid.subdomain_num_cubes_per_dim = 16;
        Parameters bad = p;
        bad.cube_size = 0.0f;
        FrameSeries series(0, 2);
        CHECK(series.depth() == 2 && series.in_flight() == 0);
        CHECK(series.submit(frames[0], p) == 0);
        CHECK(series.submit(frames[1], p) == 1);
        try {  // full
            series.submit(frames[2], p);
            CHECK(false);
        } catch (const ReconstructionError& e) {
            CHECK(e.variant == ReconstructionError::Variant::InvalidArgument);
        }
        SurfaceReconstruction out;
        for (int f = 0; f < 5; ++f) {
            if (f == 3) {  // (frame 3 was submitted with a cube size of 0: the reference panics, the series reports and continues)
                try {
                    series.next(out);
                    CHECK(false);
                } catch (const ReconstructionError& e) {
                    CHECK(e.variant == ReconstructionError::Variant::Unknown);
                }
            } else {
                CHECK(series.next(out) == (uint64_t)f);
                SurfaceReconstruction direct = ctx.reconstruct_surface(frames[f], p);
                CHECK(out.mesh.vertices == direct.mesh.vertices && out.mesh.triangles == direct.mesh.triangles);
                CHECK(out.particle_densities && *out.particle_densities == *direct.particle_densities);
                CHECK(closed_manifold(out.mesh));
            }
            if (f + 2 < 5) series.submit(frames[f + 2], f + 2 == 3 ? bad : p);
        }
        CHECK(series.in_flight() == 0);
    }
    // --- empty input is Ok with an empty mesh (SURVEY 8b edge behaviour) ---
    {
        Parameters p = Parameters::relative(0.025f, 4.0f, 1.0f);
        SurfaceReconstruction s = ctx.reconstruct_surface({}, p);
        CHECK(s.mesh.vertices.empty() && s.mesh.triangles.empty());
        CHECK(s.particle_densities && s.particle_densities->empty());
        UniformGrid g = ctx.grid_for_reconstruction({}, p);
        CHECK(g.cells_per_dim[0] == 12);
    }
    // --- errors ---
    {
        Parameters p = Parameters::relative(0.025f, 4.0f, 1.0f);
        p.cube_size = 0.0f;
        try {
            ctx.reconstruct_surface({{0.0f, 0.0f, 0.0f}}, p);
            CHECK(false);
        } catch (const ReconstructionError& e) {
            CHECK(e.variant == ReconstructionError::Variant::Unknown);  // the reference panics here
        }
        // particle AABB that excludes everything: Ok, empty mesh, flags all false
        p = Parameters::relative(0.025f, 4.0f, 1.0f);
        p.particle_aabb = Aabb3d{{1.0f, 1.0f, 1.0f}, {2.0f, 2.0f, 2.0f}};
        SurfaceReconstruction s = ctx.reconstruct_surface({{0.3f, 0.2f, 0.1f}, {0.35f, 0.2f, 0.1f}}, p);
        CHECK(s.mesh.vertices.empty());
        CHECK(s.particle_inside_aabb && s.particle_inside_aabb->size() == 2 && !(*s.particle_inside_aabb)[0]);
        CHECK(s.particle_densities && s.particle_densities->empty());
    }
    if (failures == 0) std::printf("cpp host: all checks passed\n");
    return failures == 0 ? 0 : 1;
}
