// Host-only C++ test of splashsurf::postprocessing::marching_cubes_cleanup (include/splashsurf_hip.hpp): no device needed.
// Mesh: an octahedron whose +x apex is split into two nearby vertices that snap to the same point of a unit grid and get
// collapsed (postprocessing.rs:99-242).  Checks the result of this small known case; bit-parity with the reference is pinned
// by tests/test_post.py on the reference's own meshes.
#include <cmath>
#include <cstdio>
#include <map>
#include <set>

#include "splashsurf_hip.hpp"

using namespace splashsurf;

static int failures = 0;
#define CHECK(cond)                                                                \
    do {                                                                           \
        if (!(cond)) {                                                             \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            ++failures;                                                            \
        }                                                                          \
    } while (0)

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
    // octahedron around (1.4, 1.4, 1.4), radius 0.9: its six apexes snap to six different grid points; vertex 6 sits next
    // to the +x apex (vertex 0) and snaps to the same grid point (2, 1, 1)
    TriMesh3d mesh;
    mesh.vertices = {{2.3f, 1.4f, 1.4f}, {0.5f, 1.4f, 1.4f}, {1.4f, 2.3f, 1.4f}, {1.4f, 0.5f, 1.4f}, {1.4f, 1.4f, 2.3f}, {1.4f, 1.4f, 0.5f},
                     {2.2f, 1.45f, 1.45f}};
    // octahedron with apex 0 replaced by the pair (0, 6): faces around +x use 0 or 6, plus two thin faces between them
    mesh.triangles = {{6, 2, 4}, {0, 6, 4}, {0, 4, 3}, {0, 3, 5}, {0, 5, 2}, {0, 2, 6},  // +x fan (0 and 6 share the edge 0-6)
                      {1, 4, 2}, {1, 3, 4}, {1, 5, 3}, {1, 2, 5}};
    CHECK(closed_manifold(mesh));
    UniformGrid grid;
    grid.aabb.min = {0.0f, 0.0f, 0.0f};
    grid.aabb.max = {3.0f, 3.0f, 3.0f};
    grid.cell_size = 1.0f;
    grid.points_per_dim = {4, 4, 4};
    grid.cells_per_dim = {3, 3, 3};
    TriMesh3d before = mesh;
    auto conn = postprocessing::marching_cubes_cleanup<float>(mesh, grid, std::nullopt, 5, false);
    // 6 is collapsed into 0, which moves to the average of the two; the two faces at the edge 0-6 disappear
    CHECK(mesh.vertices.size() == 6);
    CHECK(mesh.triangles.size() == 8);
    CHECK(closed_manifold(mesh));
    CHECK(conn.size() == mesh.vertices.size());
    for (const auto& ring : conn) CHECK(ring.size() == 4);
    CHECK(mesh.vertices[0][0] == (2.3f * 1.0f + 2.2f * 1.0f) / 2.0f && mesh.vertices[0][1] == (1.4f * 1.0f + 1.45f * 1.0f) / 2.0f && mesh.vertices[0][2] == mesh.vertices[0][1]);
    for (size_t i = 1; i < 6; ++i) CHECK(mesh.vertices[i] == before.vertices[i]);
    // keep_vertices: the removed vertex stays in the array (without connectivity), indices are not renumbered
    TriMesh3d kept = before;
    auto conn_kept = postprocessing::marching_cubes_cleanup<float>(kept, grid, std::nullopt, 5, true);
    CHECK(kept.vertices.size() == 7 && kept.triangles.size() == 8 && conn_kept.size() == 7 && conn_kept[6].empty());
    // a snap distance smaller than the vertices' distance to the grid point (0.3 .. 0.6 cells): nothing happens
    TriMesh3d far = before;
    postprocessing::marching_cubes_cleanup<float>(far, grid, 0.1f, 5, false);
    CHECK(far.vertices.size() == 7 && far.triangles.size() == 10);
    // f64 instantiation
    TriMesh3dT<double> md;
    for (const auto& v : before.vertices) md.vertices.push_back({(double)v[0], (double)v[1], (double)v[2]});
    md.triangles = before.triangles;
    UniformGridT<double> gd;
    gd.aabb.min = {0.0, 0.0, 0.0};
    gd.aabb.max = {3.0, 3.0, 3.0};
    gd.cell_size = 1.0;
    gd.points_per_dim = {4, 4, 4};
    gd.cells_per_dim = {3, 3, 3};
    postprocessing::marching_cubes_cleanup<double>(md, gd, std::nullopt, 5, false);
    CHECK(md.vertices.size() == 6 && md.triangles.size() == 8);
    // errors: a vertex outside of the grid
    TriMesh3d bad = before;
    bad.vertices[1][0] = -5.0f;
    bool threw = false;
    try {
        postprocessing::marching_cubes_cleanup<float>(bad, grid, std::nullopt, 5, false);
    } catch (const ReconstructionError&) {
        threw = true;
    }
    CHECK(threw);
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::printf("cleanup host test ok\n");
    return 0;
}
