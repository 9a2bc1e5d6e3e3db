"""Post-processing stages: oracle vs the reference's goldens (CPU), HIP path vs oracle and goldens (GPU).

tests/golden/post_*.npz hold the reference's own mesh (vertex / triangle order matters), its particle densities and
neighbour lists, and the outputs of the reference's functions on them (tools/gen_goldens.py --post-only)."""
import json
import os

import numpy as np
import pytest

import mesh_compare as MC
from conftest import golden_input, load_golden, device_name, device_sync

POST = ["post_cube_2366", "post_f64_cube_2366"]


def _case(name):
    g = load_golden(name)
    dt = g["vertices"].dtype.type
    prm = json.loads(str(g["params"]))
    pts = golden_input(g).astype(dt)
    return g, dt, prm, pts


def _tol(dt):
    return 2e-6 if dt == np.float32 else 1e-14


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


@pytest.mark.parametrize("name", POST)
def test_oracle_post_stages_match_reference(oracle, name):
    g, dt, prm, pts = _case(name)
    U = np.uint32 if dt == np.float32 else np.uint64
    V, T = g["vertices"], g["triangles"].astype(np.uint64)
    row, nbr = oracle.post_vertex_connectivity(V.shape[0], T)
    assert np.array_equal(row.astype(np.int64), g["conn_row_ptr"]) and np.array_equal(nbr.astype(np.int64), g["conn_neighbors"].astype(np.int64))
    # deterministic reference functions: bit-identical
    assert np.array_equal(oracle.post_laplacian_smoothing(V, row, nbr, 5, 1.0, g["weights"]).view(U), g["smoothed_5_w"].view(U))
    assert np.array_equal(oracle.post_laplacian_smoothing(V, row, nbr, 4, 0.7, np.ones(V.shape[0], dt)).view(U), g["smoothed_4_b07"].view(U))
    assert np.array_equal(oracle.post_smooth_normals(g["normals"], row, nbr, 3).view(U), g["smoothed_normals_3"].view(U))
    # sequential restatement vs the reference's parallel normals; SPH sums in another order than the R-tree's
    assert np.abs(oracle.post_vertex_normals(V, T) - g["normals"]).max() <= _tol(dt)
    h, mass, rho = dt(prm["compact_support_radius"]), dt(prm["rest_mass"]), g["densities"]
    assert np.abs(oracle.post_sph_normals(pts, rho, mass, h, V) - g["sph_normals"]).max() <= 20 * _tol(dt)
    assert _rel(oracle.post_sph_interpolate(pts, rho, mass, h, g["q"], V, False), g["sph_q"]) <= 20 * _tol(dt)
    assert _rel(oracle.post_sph_interpolate(pts, rho, mass, h, g["q"], V, True), g["sph_q_corrected"]) <= 20 * _tol(dt)
    assert _rel(oracle.post_sph_interpolate(pts, rho, mass, h, g["qv"], V, True), g["sph_v_corrected"]) <= 20 * _tol(dt)
    # smoothing weights of the reference's pipeline
    wnc = oracle.post_weighted_neighbor_counts(pts, g["nb_row_ptr"], g["nb_indices"], h)
    wnn = oracle.post_sph_interpolate(pts, rho, mass, h, wnc, g["pipe_raw_vertices"], True)
    assert _rel(wnn, g["pipe_wnn"]) <= 50 * _tol(dt)
    assert np.array_equal(oracle.post_smoothing_weights(g["pipe_wnn"], 13.0).view(U), g["pipe_sw"].view(U))


def _oracle_pipeline(oracle, pts, rho, nb_ptr, nb_idx, V, T, prm, dt, sph_normals=False):
    """The reference's recipe (reconstruct.rs:1085-1345) composed from the oracle's stage functions, on a given raw mesh."""
    h, mass = dt(prm["compact_support_radius"]), dt(prm["rest_mass"])
    row, nbr = oracle.post_vertex_connectivity(V.shape[0], T)
    wnc = oracle.post_weighted_neighbor_counts(pts, nb_ptr, nb_idx, h)
    wnn = oracle.post_sph_interpolate(pts, rho, mass, h, wnc, V, True)
    sw = oracle.post_smoothing_weights(wnn, 13.0)
    Vs = oracle.post_laplacian_smoothing(V, row, nbr, 25, 1.0, sw)
    raw_n = oracle.post_sph_normals(pts, rho, mass, h, Vs) if sph_normals else oracle.post_vertex_normals(Vs, T)
    n = oracle.post_smooth_normals(raw_n, row, nbr, 10)
    return dict(wnn=wnn, sw=sw, vertices=Vs, raw_normals=raw_n, normals=n)


@pytest.mark.parametrize("name", POST)
def test_oracle_pipeline_matches_reference_pipeline(oracle, name):
    """Same raw mesh as the reference's pipeline run: the composed recipe stays within the SPH-order tolerance."""
    g, dt, prm, pts = _case(name)
    out = _oracle_pipeline(oracle, pts, g["densities"], g["nb_row_ptr"], g["nb_indices"], g["pipe_raw_vertices"], g["triangles"].astype(np.uint64), prm, dt)
    cs = float(g["cell_size"])
    assert np.abs(out["sw"] - g["pipe_sw"]).max() <= (1e-4 if dt == np.float32 else 1e-12)
    assert np.abs(out["vertices"] - g["pipe_vertices"]).max() <= (1e-4 if dt == np.float32 else 1e-12) * cs
    assert np.abs(out["raw_normals"] - g["pipe_raw_normals"]).max() <= (2e-3 if dt == np.float32 else 1e-10)
    assert np.abs(out["normals"] - g["pipe_normals"]).max() <= (2e-3 if dt == np.float32 else 1e-10)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
def _to(x, device):
    import torch
    if device == "hbm":
        t = torch.from_numpy(np.ascontiguousarray(x))
        return t.to(device_name())
    return np.ascontiguousarray(x)


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["host", "hbm"])
@pytest.mark.parametrize("name", POST)
def test_gpu_post_stages_bit_identical_to_oracle(gpu_ctx, oracle, name, where):
    """Every ss_post_* stage against the oracle on the reference's mesh: bit-identical, with host arrays (staged by
    the library) and with HBM-resident torch tensors (used in place)."""
    from splashsurf_amd import postprocessing as PP
    g, dt, prm, pts = _case(name)
    U = np.uint32 if dt == np.float32 else np.uint64
    V, T = g["vertices"], g["triangles"].astype(np.uint64)
    h, mass, rho = dt(prm["compact_support_radius"]), dt(prm["rest_mass"]), g["densities"]
    t32 = _to(T.astype(np.int32) if where == "hbm" else T.astype(np.uint32), where)
    conn = PP.vertex_vertex_connectivity(V.shape[0], t32, gpu_ctx)
    orow, onbr = oracle.post_vertex_connectivity(V.shape[0], T)
    assert np.array_equal(_np(conn.row_ptr).astype(np.int64), orow.astype(np.int64)) and np.array_equal(_np(conn.neighbors).astype(np.int64), onbr.astype(np.int64))
    assert np.array_equal(_np(conn.row_ptr).astype(np.int64), g["conn_row_ptr"])  # = the reference's
    dV = _to(V, where)
    nrm = PP.vertex_normals(dV, t32, gpu_ctx)
    assert np.array_equal(_np(nrm).view(U), oracle.post_vertex_normals(V, T).view(U))
    assert np.abs(_np(nrm) - g["normals"]).max() <= _tol(dt)
    # smoothing (bit-identical to the reference as well)
    for iters, beta, w, key in ((5, 1.0, g["weights"], "smoothed_5_w"), (4, 0.7, np.ones(V.shape[0], dt), "smoothed_4_b07")):
        mesh = PP.TriMesh3d(_to(V.copy(), where), t32, gpu_ctx)
        PP.laplacian_smoothing_parallel(mesh, conn, iterations=iters, beta=beta, weights=_to(w, where))
        assert np.array_equal(_np(mesh.vertices).view(U), g[key].view(U)), key
    n3 = _to(g["normals"].copy(), where)
    PP.laplacian_smoothing_normals_parallel(n3, conn, iterations=3, context=gpu_ctx)
    assert np.array_equal(_np(n3).view(U), g["smoothed_normals_3"].view(U))
    # SPH interpolation: same summation order as the oracle
    interp = PP.SphInterpolator(_to(pts, where), _to(rho, where), mass, h, context=gpu_ctx)
    assert np.array_equal(_np(interp.interpolate_normals(dV)).view(U), oracle.post_sph_normals(pts, rho, mass, h, V).view(U))
    for q, fo in ((g["q"], False), (g["q"], True), (g["qv"], True)):
        got = _np(interp.interpolate_quantity(_to(q, where), dV, first_order_correction=fo))
        assert np.array_equal(got.view(U), oracle.post_sph_interpolate(pts, rho, mass, h, q, V, fo).view(U))
    assert _rel(_np(interp.interpolate_quantity(_to(g["q"], where), dV, first_order_correction=True)), g["sph_q_corrected"]) <= 20 * _tol(dt)
    # smoothing weights
    wnc = PP.weighted_neighbor_counts(_to(pts, where), _to(g["nb_row_ptr"].astype(np.int64), where), _to(g["nb_indices"].astype(np.int32), where), h, gpu_ctx)
    assert np.array_equal(_np(wnc).view(U), oracle.post_weighted_neighbor_counts(pts, g["nb_row_ptr"], g["nb_indices"], h).view(U))
    sw = PP.smoothing_weights(_to(g["pipe_wnn"], where), 13.0, gpu_ctx)
    assert np.array_equal(_np(sw).view(U), g["pipe_sw"].view(U))


@pytest.mark.gpu
@pytest.mark.parametrize("name", POST)
def test_gpu_pipeline_matches_oracle_and_reference(gpu_ctx, oracle, name):
    """reconstruction_pipeline on the GPU (mesh stays in HBM between the stages): bit-identical to the oracle's recipe
    on the same raw mesh; against the reference's own pipeline run, vertex by vertex (matched through the raw mesh's
    grid edges), within the tolerance of the SPH summation order."""
    from splashsurf_amd import postprocessing as PP
    g, dt, prm, pts = _case(name)
    U = np.uint32 if dt == np.float32 else np.uint64
    mwd, rec = PP.reconstruction_pipeline(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                          subdomain_grid=True, subdomain_grid_auto_disable=False, simd=False, mesh_smoothing_iters=25, mesh_smoothing_weights=True,
                                          mesh_smoothing_weights_normalization=13.0, compute_normals=True, sph_normals=False, normals_smoothing_iters=10,
                                          output_mesh_smoothing_weights=True, output_raw_normals=True, output_raw_mesh=True, context=gpu_ctx)
    raw_v, T = rec.mesh.vertices, rec.mesh.triangles
    assert np.array_equal(rec.particle_densities.view(U), g["densities"].view(U))
    ptr, idx = rec.particle_neighbors_csr
    out = _oracle_pipeline(oracle, pts, rec.particle_densities, ptr, idx, raw_v, T, prm, dt)
    pa = mwd.point_attributes
    assert np.array_equal(pa["wnn"].view(U), out["wnn"].view(U))
    assert np.array_equal(pa["sw"].view(U), out["sw"].view(U))
    assert np.array_equal(mwd.mesh.vertices.view(U), out["vertices"].view(U))
    assert np.array_equal(pa["raw_normals"].view(U), out["raw_normals"].view(U))
    assert np.array_equal(pa["normals"].view(U), out["normals"].view(U))
    assert np.array_equal(mwd.mesh.triangles, T)
    # reference pipeline: match vertices through the raw meshes
    ids_ref = MC.geometric_cluster_ids(g["pipe_raw_vertices"], g["grid_min"], g["cell_size"], g["n_points"])
    ids_got = MC.geometric_cluster_ids(raw_v, g["grid_min"], g["cell_size"], g["n_points"])
    order_ref, order_got = np.argsort(ids_ref, kind="stable"), np.argsort(ids_got, kind="stable")
    assert np.array_equal(ids_ref[order_ref], ids_got[order_got])
    uniq = np.concatenate([[True], np.diff(ids_ref[order_ref]) != 0]) & np.concatenate([np.diff(ids_ref[order_ref]) != 0, [True]])  # unambiguous matches only
    a, b = order_ref[uniq], order_got[uniq]
    assert uniq.mean() > 0.95
    cs = float(g["cell_size"])
    loose = dt == np.float32
    assert np.abs(pa["sw"][b] - g["pipe_sw"][a]).max() <= (1e-4 if loose else 1e-12)
    assert np.abs(mwd.mesh.vertices[b] - g["pipe_vertices"][a]).max() <= (1e-4 if loose else 1e-12) * cs
    assert np.abs(pa["normals"][b] - g["pipe_normals"][a]).max() <= (2e-3 if loose else 1e-10)
    # SPH normals variant, no smoothing
    mwd2, _ = PP.reconstruction_pipeline(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                         subdomain_grid=True, subdomain_grid_auto_disable=False, simd=False, compute_normals=True, sph_normals=True, mesh_smoothing_weights=False,
                                         context=gpu_ctx)
    assert np.abs(mwd2.point_attributes["normals"][b] - g["pipe_sph_normals"][a]).max() <= (1e-4 if loose else 1e-12)
    with pytest.raises(NotImplementedError):
        PP.reconstruction_pipeline(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.75, decimate_barnacles=True, context=gpu_ctx)


def test_clamp_with_aabb_matches_reference(oracle):
    """Mesh3d::par_clamp_with_aabb through the reference's pipeline (mesh_aabb_min/max): same vertices, same triangles,
    same order; point attributes are filtered with the vertices."""
    from splashsurf_amd.postprocessing import clamp_with_aabb
    g = load_golden("post_clamp_cube_2366")
    raw_v, raw_t = g["raw_v"], g["raw_t"].astype(np.uint64)
    lo, hi = g["aabb"]
    normals = oracle.post_vertex_normals(raw_v, raw_t)
    for clamp in (1, 0):
        v, t, attrs = clamp_with_aabb(raw_v, raw_t, lo, hi, clamp_vertices=bool(clamp), point_attributes={"normals": normals})
        assert np.array_equal(v.view(np.uint32), g["clamp%d_v" % clamp].view(np.uint32))
        assert np.array_equal(t.astype(np.int64), g["clamp%d_t" % clamp].astype(np.int64))
        assert np.abs(attrs["normals"] - g["clamp%d_normals" % clamp]).max() <= 2e-6
        if clamp:
            assert np.all(v >= lo) and np.all(v <= hi)
    v, t, _ = clamp_with_aabb(raw_v, raw_t, lo, hi, keep_vertices=True)
    assert v.shape == raw_v.shape and t.shape[0] == g["clamp1_t"].shape[0]


def test_sequential_mesh_stages_are_refused():
    """marching_cubes_cleanup (postprocessing.rs:99-242), barnacle decimation and quad conversion are sequential host stages of
    the reference and outside this library: requesting one raises before any work is done."""
    from splashsurf_amd import postprocessing as PP
    pts = np.zeros((1, 3), np.float32)
    kw = dict(particle_radius=0.025, smoothing_length=2.0, cube_size=1.0)
    with pytest.raises(NotImplementedError):
        PP.reconstruction_pipeline(pts, mesh_cleanup=True, **kw)
    with pytest.raises(NotImplementedError):
        PP.reconstruction_pipeline(pts, decimate_barnacles=True, **kw)
    assert not hasattr(PP, "marching_cubes_cleanup")


def _apply_mesh_check_mutation(T, ops):
    """The edits of tools/gen_goldens.py (apply_mesh_check_mutation), replayed on the same raw mesh."""
    for op in ops:
        if op[0] == "copy_face":
            T[op[1]] = T[op[2]]
        elif op[0] == "merge_vertices":
            (fa, ca), (where, cb) = op[1], op[2]
            fb = len(T) // 2 if where == "half" else len(T) // 3
            a, b = int(T[fa, ca]), int(T[fb, cb])
            T[T == b] = a
    return T


def test_mesh_checks_give_the_reference_messages():
    """marching_cubes::check_mesh_consistency (marching_cubes.rs:129-213): holes, non-manifold edges and non-manifold
    vertices on edited copies of a raw mesh -- the same findings, worded as the reference words them."""
    from splashsurf_amd import postprocessing as PP
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mesh_check_messages.json")))
    g = load_golden(fx["base"][:-4])
    V = g["vertices"]
    for name, ops in fx["mutations"].items():
        T = _apply_mesh_check_mutation(g["triangles"].astype(np.uint64).copy(), ops)
        for key, expect in fx["messages"][name].items():
            closed, manifold = key == "closed=1,manifold=1" or key.startswith("closed=1"), key.endswith("manifold=1")
            assert PP.check_mesh_consistency(V, T, check_closed=closed, check_manifold=manifold) == expect, (name, key)
    assert fx["messages"]["good"]["closed=1,manifold=1"] is None and "boundary edges" in fx["messages"]["duplicate_face"]["closed=1,manifold=1"]
    # orientation (reconstruct.rs:1480-1540): a consistently oriented closed mesh passes; a lone flipped face on a flat patch is found
    assert PP.check_mesh_orientation(V, g["triangles"]) is None
    quad_v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 0], [2, 1, 0]], dtype=np.float32)
    quad_t = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]], dtype=np.uint64)
    assert PP.check_mesh_orientation(quad_v, quad_t) is None
    flipped = quad_t.copy()
    flipped[3] = flipped[3][::-1]
    msg = PP.check_mesh_orientation(quad_v, flipped)
    assert msg is not None and msg.startswith("Mesh is not consistently oriented. Found ")
