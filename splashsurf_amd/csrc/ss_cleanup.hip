// ss_cleanup.hip -- HOST stage: `postprocessing::marching_cubes_cleanup` (splashsurf_lib/src/postprocessing.rs:99-242) on the
// half-edge mesh of splashsurf_lib/src/halfedge_mesh.rs.
//
// This is not a device kernel and not part of the hot path: the reference runs it as one sequential sweep over the vertices in
// index order in which every half-edge collapse changes the connectivity the next legality test sees, so its result is
// defined by that order.  It is restated here for the host so that the binary's default recipe (`--mesh-smoothing-iters`
// switches the cleanup on, splashsurf/src/reconstruct.rs:201-214) can be reproduced; the stages before and after it run on
// the MI355X.  Pinned by vectors of the reference itself (tests/golden/cleanup_*.npz, tools/gen_goldens.py --cleanup-only).
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "ss_host.h"

namespace {

// Outgoing half-edges of a vertex: valence is 4..8 almost everywhere on a marching cubes mesh, so the first ten entries live
// inline and only larger rings touch the heap (the reference's Vec<Vec<usize>> costs one allocation per vertex).
class Ring {
  public:
    Ring() = default;
    Ring(const Ring& o) { assign(o); }
    Ring& operator=(const Ring& o) {
        if (this != &o) assign(o);
        return *this;
    }
    ~Ring() { delete[] heap_; }
    size_t size() const { return n_; }
    const uint32_t* begin() const { return data(); }
    const uint32_t* end() const { return data() + n_; }
    void clear() { n_ = 0; }
    void push_back(uint32_t v) {
        if (n_ == cap_) grow();
        data()[n_++] = v;
    }
    void remove_value(uint32_t x) {  // Vec::retain(|h| *h != x)
        uint32_t* d = data();
        uint32_t w = 0;
        for (uint32_t r = 0; r < n_; ++r)
            if (d[r] != x) d[w++] = d[r];
        n_ = w;
    }

  private:
    static constexpr uint32_t kInline = 10;
    uint32_t* data() { return heap_ ? heap_ : inline_; }
    const uint32_t* data() const { return heap_ ? heap_ : inline_; }
    void assign(const Ring& o) {
        n_ = 0;
        for (uint32_t v : o) push_back(v);
    }
    void grow() {
        const uint32_t nc = cap_ * 2;
        uint32_t* h = new uint32_t[nc];
        std::memcpy(h, data(), n_ * sizeof(uint32_t));
        delete[] heap_;
        heap_ = h;
        cap_ = nc;
    }
    uint32_t inline_[kInline];
    uint32_t* heap_ = nullptr;
    uint32_t n_ = 0, cap_ = kInline;
};

struct HalfEdge {  // halfedge_mesh.rs:17-30 (`idx` is the position in the array)
    uint32_t to;
    int32_t face;  // -1: boundary
    int32_t next;  // -1: none
    uint32_t opposite;
};

template <class R>
struct HalfEdgeMesh {
    std::vector<std::array<R, 3>> vertices;
    std::vector<std::array<uint32_t, 3>> triangles;
    std::vector<HalfEdge> he;
    std::vector<Ring> vmap;  // vertex_half_edge_map
    std::vector<uint8_t> removed_v, removed_t;

    // halfedge_mesh.rs:134-147: first outgoing half-edge of `from` that points to `to`
    int64_t half_edge(uint32_t from, uint32_t to) const {
        for (uint32_t h : vmap[from])
            if (he[h].to == to) return (int64_t)h;
        return -1;
    }

    // halfedge_mesh.rs:497-556 (From<TriMesh3d>)
    void build(const R* v, uint64_t nv, const uint32_t* t, uint64_t nt) {
        vertices.resize(nv);
        for (uint64_t i = 0; i < nv; ++i) vertices[i] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
        vmap.assign(nv, Ring());
        triangles.resize(nt);
        he.reserve(nt * 3);
        for (uint64_t f = 0; f < nt; ++f) {
            const std::array<uint32_t, 3> tri = {t[3 * f], t[3 * f + 1], t[3 * f + 2]};
            triangles[f] = tri;
            uint32_t tri_hes[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) {
                const uint32_t from = tri[i], to = tri[(i + 1) % 3];
                const int64_t existing = half_edge(from, to);
                if (existing >= 0) {
                    tri_hes[i] = (uint32_t)existing;
                    he[(size_t)existing].face = (int32_t)f;
                } else {
                    const uint32_t idx = (uint32_t)he.size();
                    he.push_back(HalfEdge{to, (int32_t)f, -1, idx + 1});   // inner (counter-clockwise) edge
                    he.push_back(HalfEdge{from, -1, -1, idx});            // outer edge
                    tri_hes[i] = idx;
                    vmap[from].push_back(idx);
                    vmap[to].push_back(idx + 1);
                }
            }
            for (int i = 0; i < 3; ++i) he[tri_hes[i]].next = (int32_t)tri_hes[(i + 1) % 3];
        }
        removed_v.assign(nv, 0);
        removed_t.assign(nt, 0);
    }

    // halfedge_mesh.rs:204-256; `h` is the index of the half-edge v0 -> v1 (v0 is removed by the collapse)
    bool is_collapse_ok(uint32_t h) const {
        const HalfEdge v0v1 = he[h];
        const HalfEdge v1v0 = he[v0v1.opposite];
        const uint32_t v0 = v1v0.to, v1 = v0v1.to;
        // 0: boundary collapse, 1: faceless edge, 2: opposite vertex in `out`
        auto check_opposite_vertex = [&](const HalfEdge& e, int64_t& out) -> int {
            if (e.face < 0) return 1;
            const HalfEdge h1 = he[(size_t)e.next];   // a half-edge with a face has a successor
            const HalfEdge h2 = he[(size_t)h1.next];
            if (he[h1.opposite].face < 0 && he[h2.opposite].face < 0) return 0;
            out = h1.to;
            return 2;
        };
        int64_t v_pos = -1, v_neg = -1;
        const int r_pos = check_opposite_vertex(v0v1, v_pos);
        if (r_pos == 0) return false;
        const int r_neg = check_opposite_vertex(v1v0, v_neg);
        if (r_neg == 0) return false;
        if (r_pos == 1 || r_neg == 1) return false;  // FacelessEdge
        for (uint32_t hh : vmap[v0]) {
            const uint32_t vv = he[hh].to;
            if (vv != v1 && (int64_t)vv != v_pos && (int64_t)vv != v_neg && half_edge(vv, v1) >= 0) return false;  // IntersectionOfOneRing
        }
        return true;
    }

    // halfedge_mesh.rs:268-374
    void half_edge_collapse(uint32_t h) {
        const HalfEdge e = he[h];
        const uint32_t e_idx = h;
        const uint32_t eo_idx = e.opposite;
        const HalfEdge eo = he[eo_idx];
        const uint32_t v_from = eo.to, v_to = e.to;
        const uint32_t en_idx = (uint32_t)e.next;
        const HalfEdge en = he[en_idx];
        const uint32_t enn_idx = (uint32_t)en.next;
        const HalfEdge enn = he[enn_idx];
        const uint32_t eon_idx = (uint32_t)eo.next;
        const HalfEdge eon = he[eon_idx];
        const uint32_t eonn_idx = (uint32_t)eon.next;
        const HalfEdge eonn = he[eonn_idx];
        const uint32_t v_pos = en.to, v_neg = eon.to;

        const Ring conn_from = vmap[v_from];
        Ring conn_to = vmap[v_to];

        if (e.face >= 0) removed_t[(size_t)e.face] = 1;
        if (eo.face >= 0) removed_t[(size_t)eo.face] = 1;
        removed_v[v_from] = 1;
        // (the set of removed half-edges of the reference is never read on this path)

        if (v_pos == v_neg) {  // two opposite but coincident faces
            removed_v[v_to] = 1;
            removed_v[v_pos] = 1;
            vmap[v_from].clear();
            vmap[v_to].clear();
            vmap[v_pos].clear();
            return;
        }
        for (uint32_t hh : conn_from) {
            const int32_t f = he[hh].face;
            if (f >= 0)
                for (auto& i : triangles[(size_t)f])
                    if (i == v_from) i = v_to;
        }
        {
            const uint32_t no = en.opposite, nno = enn.opposite;
            he[no].opposite = nno;
            he[nno].opposite = no;
            const uint32_t ono = eon.opposite, onno = eonn.opposite;
            he[ono].opposite = onno;
            he[onno].opposite = ono;
        }
        conn_to.remove_value(en_idx);
        conn_to.remove_value(eo_idx);
        for (uint32_t hh : conn_from)
            if (hh != e_idx && hh != eon_idx) conn_to.push_back(hh);
        for (uint32_t hh : conn_to) {
            HalfEdge& opp = he[he[hh].opposite];
            if (opp.to == v_from) opp.to = v_to;
        }
        vmap[v_to] = conn_to;
        vmap[v_from].clear();
        vmap[v_pos].remove_value(enn_idx);
        vmap[v_neg].remove_value(eonn_idx);
    }
};

template <class R>
struct GridOf;
template <>
struct GridOf<float> { using type = ss_grid_f32; };
template <>
struct GridOf<double> { using type = ss_grid_f64; };

struct CleanupResult {
    uint64_t n_vertices = 0, n_triangles = 0, n_connectivity = 0;
};

template <class R>
ss_status cleanup_impl(ss_context* ctx, const R* vertices_in, uint64_t nv, const uint32_t* triangles_in, uint64_t nt, const typename GridOf<R>::type* grid,
                       int has_snap, R max_rel_snap_distance, uint64_t max_iter, int keep_vertices, R* vertices_out, uint32_t* triangles_out,
                       uint64_t* conn_row_out, uint32_t* conn_idx_out, uint64_t conn_capacity, uint64_t* counts_out) {
    if ((!vertices_in && nv) || (!triangles_in && nt) || !grid || !vertices_out || !triangles_out || !conn_row_out || (!conn_idx_out && conn_capacity) || !counts_out)
        return fail(ctx, SS_ERR_INVALID_ARGUMENT, "marching_cubes_cleanup: null argument");
    if (nv >= (1ull << 32) || nt * 3 >= (1ull << 32)) return fail(ctx, SS_ERR_UNSUPPORTED, "marching_cubes_cleanup: mesh too large for 32-bit indices");
    for (uint64_t i = 0; i < nt * 3; ++i)
        if (triangles_in[i] >= nv) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "marching_cubes_cleanup: triangle index out of range");
    const R cs = grid->cell_size;
    const R half_dx = cs / (R(1.0) + R(1.0));
    const R snap = max_rel_snap_distance * cs;
    const R max_snap_sq = snap * snap;  // powi(2)
    const int64_t np[3] = {grid->n_points[0], grid->n_points[1], grid->n_points[2]};

    // nearest grid point of every vertex (postprocessing.rs:112-155); MC vertices lie inside the grid
    std::vector<int64_t> nearest(nv);
    auto point_coord = [&](const int64_t ijk[3], R out[3]) {  // uniform_grid.rs:418-431
        for (int d = 0; d < 3; ++d) out[d] = grid->aabb_min[d] + (R)ijk[d] * cs;
    };
    for (uint64_t i = 0; i < nv; ++i) {
        int64_t ijk[3];
        R mc[3];
        for (int d = 0; d < 3; ++d) {
            const R normalized = (vertices_in[3 * i + d] - grid->aabb_min[d]) / cs;  // uniform_grid.rs:444-451
            ijk[d] = (int64_t)std::floor(normalized);
            if (ijk[d] < 0 || ijk[d] >= np[d]) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "marching_cubes_cleanup: vertex outside of the grid");
        }
        point_coord(ijk, mc);
        for (int d = 0; d < 3; ++d)
            if ((vertices_in[3 * i + d] - mc[d]) > half_dx) {
                if (ijk[d] == np[d] - 1) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "marching_cubes_cleanup: vertex outside of the grid");
                ijk[d] += 1;
            }
        nearest[i] = (ijk[0] * np[1] + ijk[1]) * np[2] + ijk[2];  // uniform_grid.rs:342-345 (same value, i*np1*np2 + j*np2 + k)
    }

    HalfEdgeMesh<R> mesh;
    mesh.build(vertices_in, nv, triangles_in, nt);
    std::vector<uint64_t> sum_count(nv, 1);
    std::vector<uint32_t> buffer;
    for (uint64_t it = 0; it < max_iter; ++it) {
        uint64_t collapse_count = 0;
        for (uint32_t v0 = 0; v0 < (uint32_t)nv; ++v0) {
            if (mesh.removed_v[v0]) continue;
            int64_t ijk[3];
            {  // try_unflatten_point_index (uniform_grid.rs:380-395): always a point of the grid here
                const int64_t f = nearest[v0];
                ijk[0] = f / (np[1] * np[2]);
                ijk[1] = (f - ijk[0] * np[1] * np[2]) / np[2];
                ijk[2] = f - ijk[0] * np[1] * np[2] - ijk[1] * np[2];
            }
            R gp[3];
            point_coord(ijk, gp);
            auto within_snap = [&](uint32_t v) {
                const R dx = mesh.vertices[v][0] - gp[0], dy = mesh.vertices[v][1] - gp[1], dz = mesh.vertices[v][2] - gp[2];
                return (dx * dx + dy * dy + dz * dz) <= max_snap_sq;
            };
            if (has_snap) {
                if (within_snap(v0))
                    for (uint32_t h : mesh.vmap[v0]) {
                        const uint32_t v1 = mesh.he[h].to;
                        if (nearest[v0] == nearest[v1] && within_snap(v1)) buffer.push_back(v1);
                    }
            } else {
                for (uint32_t h : mesh.vmap[v0]) {
                    const uint32_t v1 = mesh.he[h].to;
                    if (nearest[v0] == nearest[v1]) buffer.push_back(v1);
                }
            }
            for (uint32_t v1 : buffer) {
                if (mesh.removed_v[v1]) continue;
                const int64_t h = mesh.half_edge(v1, v0);
                if (h < 0) continue;
                if (!mesh.is_collapse_ok((uint32_t)h)) continue;
                mesh.half_edge_collapse((uint32_t)h);
                ++collapse_count;
                // move to the averaged position (postprocessing.rs:209-221)
                const uint64_t n0 = sum_count[v0], n1 = sum_count[v1], nn = n0 + n1;
                for (int d = 0; d < 3; ++d) {
                    const R a = mesh.vertices[v0][d] * (R)n0;
                    const R b = mesh.vertices[v1][d] * (R)n1;
                    mesh.vertices[v0][d] = (a + b) / (R)nn;
                }
                sum_count[v0] = nn;
            }
            buffer.clear();
        }
        if (collapse_count == 0) break;
    }

    // into_parts (halfedge_mesh.rs:92-100, 433-494): connectivity = targets of the remaining half-edges, removed faces and
    // (unless keep_vertices) removed vertices filtered out, indices renumbered in order
    std::vector<uint32_t> new_index(nv, 0);
    uint64_t n_out_v = 0;
    if (keep_vertices) {
        for (uint64_t i = 0; i < nv; ++i) new_index[i] = (uint32_t)i;
        n_out_v = nv;
    } else {
        for (uint64_t i = 0; i < nv; ++i)
            if (!mesh.removed_v[i]) new_index[i] = (uint32_t)n_out_v++;
    }
    uint64_t n_out_t = 0, n_conn = 0;
    for (uint64_t f = 0; f < nt; ++f)
        if (!mesh.removed_t[f]) {
            for (int c = 0; c < 3; ++c) triangles_out[3 * n_out_t + c] = new_index[mesh.triangles[f][c]];
            ++n_out_t;
        }
    uint64_t row = 0;
    for (uint64_t i = 0; i < nv; ++i) {
        if (!keep_vertices && mesh.removed_v[i]) continue;
        for (int d = 0; d < 3; ++d) vertices_out[3 * row + d] = mesh.vertices[i][d];
        conn_row_out[row] = n_conn;
        for (uint32_t h : mesh.vmap[i]) {
            if (n_conn >= conn_capacity) return fail(ctx, SS_ERR_INVALID_ARGUMENT, "marching_cubes_cleanup: connectivity buffer too small (6 x triangles entries suffice)");
            conn_idx_out[n_conn++] = new_index[mesh.he[h].to];
        }
        ++row;
    }
    conn_row_out[row] = n_conn;
    counts_out[0] = n_out_v;
    counts_out[1] = n_out_t;
    counts_out[2] = n_conn;
    return SS_OK;
}

}  // namespace

extern "C" {

ss_status ss_post_marching_cubes_cleanup_f32(ss_context* ctx, const float* vertices, uint64_t n_vertices, const uint32_t* triangles, uint64_t n_triangles,
                                             const ss_grid_f32* grid, int has_max_rel_snap_distance, float max_rel_snap_distance, uint64_t max_iter,
                                             int keep_vertices, float* vertices_out, uint32_t* triangles_out, uint64_t* connectivity_row_out,
                                             uint32_t* connectivity_idx_out, uint64_t connectivity_capacity, uint64_t* counts_out) {
    return cleanup_impl<float>(ctx, vertices, n_vertices, triangles, n_triangles, grid, has_max_rel_snap_distance, max_rel_snap_distance, max_iter, keep_vertices,
                               vertices_out, triangles_out, connectivity_row_out, connectivity_idx_out, connectivity_capacity, counts_out);
}

ss_status ss_post_marching_cubes_cleanup_f64(ss_context* ctx, const double* vertices, uint64_t n_vertices, const uint32_t* triangles, uint64_t n_triangles,
                                             const ss_grid_f64* grid, int has_max_rel_snap_distance, double max_rel_snap_distance, uint64_t max_iter,
                                             int keep_vertices, double* vertices_out, uint32_t* triangles_out, uint64_t* connectivity_row_out,
                                             uint32_t* connectivity_idx_out, uint64_t connectivity_capacity, uint64_t* counts_out) {
    return cleanup_impl<double>(ctx, vertices, n_vertices, triangles, n_triangles, grid, has_max_rel_snap_distance, max_rel_snap_distance, max_iter, keep_vertices,
                                vertices_out, triangles_out, connectivity_row_out, connectivity_idx_out, connectivity_capacity, counts_out);
}

}  // extern "C"
