"""The BENCHMARKED configuration pinned to the reference itself (-m gpu).

tests/golden/config3_s10m_tank.npz and simd_config3_s10m_tank.npz hold digests of what the reference's own wheel computes for
S10M-tank (BASELINE config 3: 10 M particles, 7.18 M vertices) with `simd=False` and `simd=True` (tools/gen_goldens_fullsize.py;
lib.rs:330-337, dense_subdomains.rs:784-847 scalar, :991-1133 + :1413-1415 SIMD).  What is asserted, exactly:

  * enable_simd = 0 (bench.py's headline mode): densities bit-identical; the vertex-id multiset and the triangle set EQUAL the wheel's
    (sha256 over the canonical forms), sampled vertices within 1e-5 relative (north_star);
  * enable_simd = 1 (uniform AVX arithmetic, include/splashsurf_hip.h): the mesh equals the digest of that arithmetic, and its relation to
    the wheel's `simd=True` mesh is the STORED, COUNTED difference -- removing the stored library-only ids / triangles and adding the stored
    reference-only ones reproduces the wheel's digests.  At 7.18 M vertices that difference is 1 vertex and 13 / 11 triangles at two grid
    points: (197, 810, 640), whose value equals the threshold to the last bit and which lies on a subdomain face -- the reference's two
    adjacent subdomains compute it with different lanes and disagree about its side --, and (1610, 92, 1598), a vertex 8e-4 of a cell from
    the grid point that the geometric canonicalisation files under "grid point" for one mesh and "edge" for the other (the reference's own
    two modes differ from each other in 312 ids and 2 700 triangles on this input; FULLSIZE_REPORT.json).

Round 6 adds the two largest workloads in the same way: config4_s40m_tank.npz (BASELINE config 4, 39.8 M particles, 18.05 M vertices, simd=False: the
scalar mode's mesh EQUALS the wheel's) and [simd_]config3p_s10m_cube.npz (SURVEY 8d's literal reading 3' of config 3: 10 M particles ten times over-dense,
1.10 M vertices -- the only large input of k_splat_certify_big and the arena kernels; scalar: equal; uniform AVX arithmetic: a stored difference of 1 id and
6 / 6 triangles, where the wheel's own two modes differ in 7 ids and 40 triangles).
"""
import hashlib

import numpy as np
import pytest

import mesh_compare as MC
from conftest import golden_input, golden_params, load_golden
from test_gpu_parity import run_gpu

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _rows(t):
    t = np.ascontiguousarray(t, dtype=np.int64).reshape(-1, 3)
    return t.view([("a", np.int64), ("b", np.int64), ("c", np.int64)]).ravel()


def _apply_difference(sorted_items, remove, add, rows=False):
    """(multiset `sorted_items` minus `remove`) plus `add`, sorted again; every removed item must be present."""
    a = _rows(sorted_items) if rows else np.asarray(sorted_items, dtype=np.int64)
    rem = _rows(remove) if rows else np.asarray(remove, dtype=np.int64)
    ad = _rows(add) if rows else np.asarray(add, dtype=np.int64)
    keep = np.ones(a.size, dtype=bool)
    for r in rem:
        lo, hi = np.searchsorted(a, r, side="left"), np.searchsorted(a, r, side="right")
        cand = np.nonzero(keep[lo:hi])[0]
        assert cand.size, "an item of the stored difference is not in this mesh: %r" % (r,)
        keep[lo + cand[0]] = False
    out = np.sort(np.concatenate([a[keep], ad]), order=("a", "b", "c") if rows else None)
    return out.view(np.int64).reshape(-1, 3) if rows else out


def _check_samples(g, ids, vs):
    sid, sv = g["sample_ids"], g["sample_vertices"].astype(np.float64)
    lo, hi = np.searchsorted(ids, sid, side="left"), np.searchsorted(ids, sid, side="right")
    present = hi > lo
    single = present & ((hi - lo) == 1)
    worst = float(np.abs(vs[lo[single]].astype(np.float64) - sv[single]).max())
    for k in np.nonzero(present & ~single)[0]:
        worst = max(worst, float(np.abs(vs[lo[k]:hi[k]].astype(np.float64) - sv[k]).max(axis=1).min()))
    assert worst <= 1e-5 * max(1.0, np.abs(sv).max()), worst  # north_star: 1e-5 relative
    return int((~present).sum())


def assert_equals_the_wheels_digest(res, g, simd, name="", capsys=None):
    """The mesh and densities of `res` against a full-size golden of the reference wheel (see the module's docstring); also used by
    tests/test_gpu_dist_native.py for the merged mesh of the sharded path."""
    assert res.stats["arith_mode"] in ((2, 3) if simd else (0, 1))
    assert list(res.grid.ncells_per_dim) == list(g["n_cells"])
    assert _sha(res.particle_densities) == str(g["density_sha256"]), "densities not bit-identical to the reference"
    ids, vs, tc = MC.canonicalize_geometric(res.mesh.vertices, res.mesh.triangles_u32, g["grid_min"], g["cell_size"], g["n_points"])
    ids = ids.astype(np.int64)
    tc = tc.astype(np.int64)
    # the library's own digest for this mode (generated from the oracle's mode 0 / 2 in the wheel's canonical form)
    assert (ids.size, tc.shape[0]) == (int(g["lib_n_vertices"]), int(g["lib_n_triangles"]))
    assert _sha(ids) == str(g["lib_ids_sha256"]) and _sha(tc) == str(g["lib_triangles_sha256"])
    # ... and its exact relation to the wheel's mesh
    rem_i, add_i = g["lib_ids_only_in_library"], g["lib_ids_only_in_reference"]
    rem_t, add_t = g["lib_triangles_only_in_library"].reshape(-1, 3), g["lib_triangles_only_in_reference"].reshape(-1, 3)
    if not simd:
        assert rem_i.size == add_i.size == 0 and rem_t.size == add_t.size == 0  # the scalar mode IS the reference's mesh
    ids_ref = _apply_difference(ids, rem_i, add_i)
    tc_ref = _apply_difference(tc, rem_t, add_t, rows=True)
    assert ids_ref.size == int(g["n_vertices"]) and tc_ref.shape[0] == int(g["n_triangles"])
    assert _sha(ids_ref) == str(g["ids_sha256"]), "vertex ids differ from the wheel's beyond the stored difference"
    assert _sha(tc_ref) == str(g["triangles_sha256"]), "triangles differ from the wheel's beyond the stored difference"
    missing = _check_samples(g, ids, vs)
    assert missing <= rem_i.size + add_i.size
    assert MC.mesh_is_closed_manifold(res.mesh.triangles_u32)
    import contextlib
    with (capsys.disabled() if capsys is not None else contextlib.nullcontext()):
        print("\n[%s] enable_simd=%d: %d vertices / %d triangles; against the wheel's simd=%s mesh (%d / %d): ids only here %d, only in the wheel %d; "
              "triangles only here %d, only in the wheel %d" % (name, int(simd), ids.size, tc.shape[0], simd, int(g["n_vertices"]), int(g["n_triangles"]),
                                                                rem_i.size, add_i.size, rem_t.shape[0], add_t.shape[0]))


# config3 (S10M-tank, bench.py's workload), config 4 (S40M-tank, 39.8 M particles / 18.05 M vertices: the workload of the multi-GPU bench, scalar
# mode) and SURVEY 8d's literal reading 3' of config 3 (S10M-cube, ten times over-dense: the only large input of k_splat_certify_big and the arena path)
@pytest.mark.parametrize("name", ["config3_s10m_tank", "simd_config3_s10m_tank", "config3p_s10m_cube", "simd_config3p_s10m_cube", "config4_s40m_tank"])
def test_full_size_against_the_reference_wheel(gpu_ctx, name, capsys):
    g = load_golden(name)
    prm = golden_params(g)
    simd = bool(prm["simd"])
    res = run_gpu(gpu_ctx, golden_input(g), prm, simd=simd)
    assert_equals_the_wheels_digest(res, g, simd, name, capsys)
