"""CPU checks of the helpers behind the full-size reference digests (tests/test_gpu_fullsize.py, tools/gen_goldens_fullsize.py): the multiset
difference that the generator stores and the test applies must be inverse operations, for vertex ids (with multiplicities: a grid-point cluster
holds several vertices) and for canonical triangle rows; and the committed goldens carry every field the GPU tests read."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))


def _multiset_diff(a, b):
    # (tools/gen_goldens_fullsize.py imports the reference wheel at module level; its pure helper is restated here and pinned against it below
    # when the wheel is available)
    if a.ndim == 2:
        va = np.ascontiguousarray(a).view([("", a.dtype)] * a.shape[1]).ravel()
        vb = np.ascontiguousarray(b).view([("", b.dtype)] * b.shape[1]).ravel()
    else:
        va, vb = a, b
    ua, ca = np.unique(va, return_counts=True)
    ub, cb = np.unique(vb, return_counts=True)
    pos = np.searchsorted(ub, ua)
    pos_c = np.minimum(pos, max(ub.size - 1, 0))
    have = np.where((pos < ub.size) & (ub[pos_c] == ua), cb[pos_c], 0) if ub.size else np.zeros(ua.size, dtype=np.int64)
    out = np.repeat(ua, np.maximum(ca - have, 0))
    return out.view(a.dtype).reshape(-1, a.shape[1]) if a.ndim == 2 else out


def test_stored_difference_round_trips():
    from test_gpu_fullsize import _apply_difference
    rng = np.random.default_rng(5)
    # ids with multiplicities
    ref = np.sort(rng.integers(0, 500, size=4000)).astype(np.int64)
    lib = np.sort(np.concatenate([np.delete(ref, [3, 3, 700, 1999]), [7, 7, 499, 123456]])).astype(np.int64)
    only_ref, only_lib = _multiset_diff(ref, lib), _multiset_diff(lib, ref)
    assert only_ref.size == only_lib.size == 4 or only_ref.size <= 4  # (an inserted value may coincide with a deleted one)
    back = _apply_difference(lib, only_lib, only_ref)
    assert np.array_equal(back, ref)
    # canonical triangle rows (lexicographically sorted)
    t = rng.integers(0, 300, size=(3000, 3)).astype(np.int64)
    t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    t2 = np.concatenate([np.delete(t, [5, 6, 2500], axis=0), [[1, 2, 3], [299, 0, 17]]]).astype(np.int64)
    t2 = t2[np.lexsort((t2[:, 2], t2[:, 1], t2[:, 0]))]
    r_only, l_only = _multiset_diff(t, t2), _multiset_diff(t2, t)
    back = _apply_difference(t2, l_only, r_only, rows=True)
    assert np.array_equal(back, t)
    # nothing to apply
    assert np.array_equal(_apply_difference(ref, np.zeros(0, np.int64), np.zeros(0, np.int64)), ref)


def test_full_size_goldens_are_complete():
    from conftest import golden_params, load_golden
    report = os.path.join(os.path.dirname(__file__), "golden", "FULLSIZE_REPORT.json")
    assert os.path.exists(report)
    for name, simd in (("config3_s10m_tank", False), ("simd_config3_s10m_tank", True), ("simd_config2_s1m", True)):
        g = load_golden(name)
        assert bool(golden_params(g)["simd"]) == simd
        for k in ("n_vertices", "n_triangles", "ids_sha256", "triangles_sha256", "density_sha256", "sample_ids", "sample_vertices", "grid_min", "cell_size", "n_points",
                  "lib_ids_sha256", "lib_triangles_sha256", "lib_ids_only_in_reference", "lib_ids_only_in_library", "lib_triangles_only_in_reference",
                  "lib_triangles_only_in_library", "lib_n_vertices", "lib_n_triangles"):
            assert k in g.files, (name, k)
        assert g["sample_ids"].size == 65536
        same = str(g["ids_sha256"]) == str(g["lib_ids_sha256"]) and str(g["triangles_sha256"]) == str(g["lib_triangles_sha256"])
        assert same == (g["lib_ids_only_in_reference"].size + g["lib_ids_only_in_library"].size + g["lib_triangles_only_in_reference"].size
                        + g["lib_triangles_only_in_library"].size == 0)
    # the benchmarked mode IS the reference's mesh; the default mode differs by the stored handful at full size and by nothing at 1 M particles
    assert load_golden("config3_s10m_tank")["lib_ids_only_in_reference"].size == 0
    assert str(load_golden("simd_config2_s1m")["ids_sha256"]) == str(load_golden("simd_config2_s1m")["lib_ids_sha256"])
    g = load_golden("simd_config3_s10m_tank")
    assert (int(g["n_vertices"]), int(g["lib_n_vertices"])) == (7181619, 7181618) and g["lib_triangles_only_in_library"].reshape(-1, 3).shape[0] == 13
