#!/usr/bin/env python3
"""Full-size goldens: the BENCHMARKED configurations pinned to the reference itself.

Build container only: imports the reference's pre-built wheel (tools/oracle_env.py) and runs
`pysplashsurf.reconstruct_surface(..., subdomain_grid=True, subdomain_grid_auto_disable=False)` with `simd=False`
(density_grid_loop_scalar, dense_subdomains.rs:784-847) and `simd=True` (the AVX2+FMA loop with its remainder lanes and
the scalar loop for sparse subdomains, dense_subdomains.rs:991-1133, :1413-1415; lib.rs:330-337) on

  * S10M-tank  (BASELINE config 3, bench.py's default workload: 10 M particles, ~7.2 M vertices), and
  * S1M        (BASELINE config 2, simd=True; the scalar digest is tests/golden/config2_s1m.npz),
  * S40M-tank  (BASELINE config 4, 39.8 M particles, simd=False) and S10M-cube (SURVEY 8d's literal reading 3' of config 3, both flags).

Stored per case (data only): counts, sha256 of the sorted geometric vertex ids / the canonical triangle list over those
ids / the densities, and 65 536 sampled (id, vertex) pairs.  The oracle (mode 0 against simd=False; modes 1 and 2 against
simd=True) is compared with the wheel on the way; where a vertex-id set differs, the symmetric difference is COUNTED AND
STORED as multisets (vertex ids and canonical triangle rows only in the wheel's mesh / only in the other mesh), so that the GPU tests
assert the exact relation instead of a tolerance.  Report: tests/golden/FULLSIZE_REPORT.json.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_env import pysplashsurf  # noqa: E402
from oracle import oracle as O  # noqa: E402
import mesh_compare as MC  # noqa: E402
from splashsurf_amd import workloads as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_SAMPLE = 65536


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_run(p, r, l, c, t, simd):
    t0 = time.time()
    res = pysplashsurf.reconstruct_surface(p, particle_radius=r, smoothing_length=l, cube_size=c, iso_surface_threshold=t, simd=simd,
                                           multi_threading=True, subdomain_grid=True, subdomain_grid_auto_disable=False,
                                           subdomain_num_cubes_per_dim=64)
    dt = time.time() - t0
    return dict(vertices=np.asarray(res.mesh.vertices, dtype=np.float32).reshape(-1, 3),
                triangles=np.asarray(res.mesh.triangles).astype(np.int64).reshape(-1, 3),
                densities=np.asarray(res.particle_densities, dtype=np.float32),
                grid_min=np.asarray(res.grid.aabb.min, dtype=np.float64).astype(np.float32),
                cell_size=np.float32(res.grid.cell_size), n_points=np.asarray(res.grid.npoints_per_dim, dtype=np.int64),
                n_cells=np.asarray(res.grid.ncells_per_dim, dtype=np.int64), seconds=dt)


def canon(m, g):
    return MC.canonicalize_geometric(m["vertices"], m["triangles"], g["grid_min"], g["cell_size"], g["n_points"])


def multiset_diff(a, b):
    """Elements (with multiplicity) of the sorted 1-D array / lexicographically sorted row array `a` that `b` does not hold."""
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
    extra = np.maximum(ca - have, 0)
    out = np.repeat(ua, extra)
    return out.view(a.dtype).reshape(-1, a.shape[1]) if a.ndim == 2 else out


def relation(ref_c, oth_c):
    """Exact relation of two canonical meshes: MULTISET differences of the vertex ids (a grid-point cluster holds several vertices) and of
    the canonical triangle rows, and the largest relative coordinate difference over the ids both hold exactly once."""
    ia, va, ta = ref_c
    ib, vb, tb = oth_c
    only_a, only_b = multiset_diff(ia, ib), multiset_diff(ib, ia)
    t_only_a, t_only_b = multiset_diff(ta, tb), multiset_diff(tb, ta)
    ua, ca = np.unique(ia, return_counts=True)
    ub, cb = np.unique(ib, return_counts=True)
    common = np.intersect1d(ua[ca == 1], ub[cb == 1])
    pa, pb = np.searchsorted(ia, common), np.searchsorted(ib, common)
    A, B = va[pa].astype(np.float64), vb[pb].astype(np.float64)
    d = np.abs(A - B).max(axis=1) if common.size else np.zeros(0)
    rel = float((d / np.maximum(np.abs(A).max(axis=1), 1e-30)).max()) if common.size else 0.0
    nbit = int(np.all(va[pa] == vb[pb], axis=1).sum()) if common.size else 0
    rec = dict(ids_equal=bool(np.array_equal(ia, ib)), triangles_equal=bool(ta.shape == tb.shape and np.array_equal(ta, tb)),
               n_ids_only_in_reference=int(only_a.size), n_ids_only_in_other=int(only_b.size),
               n_triangles_only_in_reference=int(t_only_a.shape[0]), n_triangles_only_in_other=int(t_only_b.shape[0]),
               n_vertices=(int(ia.size), int(ib.size)), n_triangles=(int(ta.shape[0]), int(tb.shape[0])), max_rel_diff_common=rel,
               n_common_single=int(common.size), n_common_bit_equal=nbit)
    return rec, (only_a, only_b, t_only_a, t_only_b)


def store(name, ref, refc, prm, desc, extra=None):
    ids, vs, tc = refc
    sel = np.sort(np.random.default_rng(7).choice(ids.size, size=min(N_SAMPLE, ids.size), replace=False))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), n_vertices=np.int64(ids.size), n_triangles=np.int64(tc.shape[0]),
                        ids_sha256=np.array(sha(ids.astype(np.int64))), triangles_sha256=np.array(sha(tc.astype(np.int64))),
                        density_sha256=np.array(sha(ref["densities"])), sample_ids=ids[sel].astype(np.int64),
                        sample_vertices=vs[sel].astype(np.float32), grid_min=ref["grid_min"], cell_size=ref["cell_size"],
                        n_points=ref["n_points"], n_cells=ref["n_cells"], params=np.array(json.dumps(prm)), input=np.array(json.dumps(desc)),
                        **(extra or {}))


def main():
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))
    only_simd = "--simd-only" in sys.argv
    only_scalar = "--scalar-only" in sys.argv
    cases = [
        # workload, simd flags to generate, oracle modes to compare per flag
        ("s1m", dict(kind="workload", name="uniform_cube", n=1_000_000, seed=12345), {True: (1, 2)}),
        ("s10m_tank", dict(kind="workload", name="tank", scale=1.0), {False: (0,), True: (1, 2)}),
        # round 6: the two largest workloads -- config 4 (39.8 M particles, scalar mode = what the sharded bench
        # measures) and the literal reading 3' of config 3 (10 x over-dense unit cube: the only large input of the over-dense kernels)
        ("s40m_tank", dict(kind="workload", name="tank", scale=4.0 ** (1.0 / 3.0)), {False: (0,)}),
        ("s10m_cube", dict(kind="workload", name="uniform_cube", n=10_000_000, seed=12346), {False: (0,), True: (1, 2)}),
    ]
    golden_name = {"s1m": "config2_s1m", "s10m_tank": "config3_s10m_tank", "s40m_tank": "config4_s40m_tank", "s10m_cube": "config3p_s10m_cube"}
    rpath = os.path.join(GOLD, "FULLSIZE_REPORT.json")
    report = json.load(open(rpath)) if os.path.exists(rpath) else {}
    for wname, desc, flags in cases:
        if only and wname not in only:
            continue
        wl = W.WORKLOADS[wname]
        pts = np.ascontiguousarray(wl["gen"](), dtype=np.float32)
        r, l, c, t = wl["particle_radius"], wl["smoothing_length"], wl["cube_size"], 0.6
        rho_sha = None
        canon_by_flag = {}
        for simd, modes in flags.items():
            if (only_simd and not simd) or (only_scalar and simd):
                continue
            ref = ref_run(pts, r, l, c, t, simd)
            rc = canon(ref, ref)
            canon_by_flag[simd] = rc
            gname = ("simd_" if simd else "") + golden_name[wname]
            rec = dict(reference_seconds=round(ref["seconds"], 2), n_vertices=int(rc[0].size), n_triangles=int(rc[2].shape[0]))
            if rho_sha is None:
                rho_sha = sha(ref["densities"])
            assert sha(ref["densities"]) == rho_sha, "the reference's densities depend on simd"
            extra = {}
            for mode in modes:
                t0 = time.time()
                orc = O.reconstruct_surface(pts, O.make_params_relative(r, l, c, iso_surface_threshold=t, subdomain_num_cubes_per_dim=64, simd=mode))
                dt = time.time() - t0
                assert np.array_equal(ref["densities"].view(np.uint32), orc.particle_densities.view(np.uint32)), (gname, mode, "rho")
                oc = canon(dict(vertices=orc.vertices, triangles=orc.triangles), ref)
                rel, (only_ref, only_orc, t_only_ref, t_only_orc) = relation(rc, oc)
                if not rel["ids_equal"]:  # where: the grid points (i, j, k, axis; axis 3 = a cluster of vertices sitting ON the grid point) of the differing ids
                    npnt = [int(x) for x in ref["n_points"]]
                    dec = lambda i: [int((i // 4) // (npnt[1] * npnt[2])), int(((i // 4) // npnt[2]) % npnt[1]), int((i // 4) % npnt[2]), int(i % 4)]  # noqa: E731
                    rel["ids_only_in_reference_at"] = [dec(int(i)) for i in only_ref[:16]]
                    rel["ids_only_in_other_at"] = [dec(int(i)) for i in only_orc[:16]]
                rel["oracle_seconds"] = round(dt, 1)
                rec["oracle_mode_%d" % mode] = rel
                print(gname, "oracle mode", mode, json.dumps(rel), flush=True)
                if mode in (0, 2):
                    # what the HIP library computes for this flag (enable_simd 0 -> mode 0, 1 -> mode 2): its exact relation to
                    # the wheel's mesh is part of the golden
                    extra = dict(lib_ids_only_in_reference=only_ref.astype(np.int64), lib_ids_only_in_library=only_orc.astype(np.int64),
                                 lib_triangles_only_in_reference=t_only_ref.astype(np.int64).reshape(-1, 3),
                                 lib_triangles_only_in_library=t_only_orc.astype(np.int64).reshape(-1, 3),
                                 lib_n_vertices=np.int64(oc[0].size), lib_n_triangles=np.int64(oc[2].shape[0]),
                                 lib_ids_sha256=np.array(sha(oc[0].astype(np.int64))), lib_triangles_sha256=np.array(sha(oc[2].astype(np.int64))))
                del orc, oc
            prm = dict(particle_radius=r, smoothing_length=l, cube_size=c, iso_surface_threshold=t, subdomain_num_cubes_per_dim=64, simd=bool(simd))
            store(gname, ref, rc, prm, desc, extra)
            report[gname] = rec
            json.dump(report, open(rpath, "w"), indent=1, sort_keys=True)
            del ref
        if len(canon_by_flag) == 2:
            rel, _ = relation(canon_by_flag[True], canon_by_flag[False])
            report[wname + "_reference_simd_vs_scalar"] = rel
            print(wname, "reference simd vs scalar", json.dumps(rel), flush=True)
            json.dump(report, open(rpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
