#!/usr/bin/env python3
"""Build container only: the randomised configurations of tests/test_gpu_fuzz.py, oracle vs the REFERENCE (wheel).
Densities and neighbour lists must be bit-identical, meshes identical under the geometric canonicalisation
(vertex coordinates: bit-identical for the global strategy, <= 1 ulp on subdomain faces for the grid strategy).
Writes tests/golden/FUZZ_REPORT.json (a record, not a fixture)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_env import pysplashsurf  # noqa: E402
from oracle import oracle as O  # noqa: E402
import mesh_compare as MC  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

report = []
for case in F.CASES:
    dt = np.float64 if case["f64"] else np.float32
    U = np.uint64 if case["f64"] else np.uint32
    rng = np.random.default_rng(case["seed"])
    pts = (F._cloud(rng, case["n"], case["kind"], 2.0 * case["r"]) + case["offset"]).astype(np.float32).astype(dt)
    kw, okw = {}, {}
    if case["aabb"]:
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        a, b = lo + 0.2 * (hi - lo), hi - 0.1 * (hi - lo)
        kw = dict(aabb_min=[float(x) for x in a], aabb_max=[float(x) for x in b])
        okw = dict(aabb_min=np.asarray(kw["aabb_min"], dt), aabb_max=np.asarray(kw["aabb_max"], dt))
    glob = case["strategy"] == "global"
    ref = pysplashsurf.reconstruct_surface(pts, particle_radius=case["r"], rest_density=case["rest_density"], smoothing_length=case["l"], cube_size=case["c"],
                                           iso_surface_threshold=case["t"], simd=False, multi_threading=not glob, subdomain_grid=not glob,
                                           subdomain_grid_auto_disable=False, subdomain_num_cubes_per_dim=case["n_cubes"], global_neighborhood_list=True, **kw)
    par = O.make_params_relative(case["r"], case["l"], case["c"], iso_surface_threshold=case["t"], rest_density=case["rest_density"],
                                 subdomain_num_cubes_per_dim=case["n_cubes"], global_neighborhood_list=True, dtype=dt, subdomain_grid=not glob, **okw)
    orc = O.reconstruct_surface(pts, par)
    rd = np.asarray(ref.particle_densities, dtype=dt)
    assert np.array_equal(rd.view(U), orc.particle_densities.view(U)), case
    lists = ref.particle_neighbors.get_neighborhood_lists()
    ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    idx = np.concatenate([np.asarray(x, dtype=np.uint64) for x in lists]) if ptr[-1] else np.zeros(0, np.uint64)
    assert np.array_equal(ptr, orc.neighbor_ptr) and np.array_equal(idx, orc.neighbors), case
    rv = np.asarray(ref.mesh.vertices, dtype=dt).reshape(-1, 3)
    rt = np.asarray(ref.mesh.triangles).astype(np.int64).reshape(-1, 3)
    assert list(ref.grid.ncells_per_dim) == list(orc.grid["n_cells"]), case
    entry = dict(case={k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in case.items()}, n_vertices=int(rv.shape[0]))
    if rv.shape[0]:
        cmp = MC.compare_geometric(rv, rt, orc.vertices, orc.triangles, np.asarray(ref.grid.aabb.min, dtype=dt), dt(ref.grid.cell_size), ref.grid.npoints_per_dim)
        assert cmp["ids_equal"] and cmp["triangles_equal"], (case, cmp)
        assert cmp["max_rel_diff"] <= (0.0 if glob else (1e-6 if dt == np.float32 else 1e-14)), (case, cmp)
        entry["max_rel_diff"] = cmp["max_rel_diff"]
    else:
        assert orc.vertices.shape[0] == 0
    report.append(entry)
    print(len(report), entry["case"]["kind"], entry["case"]["strategy"], "R=%d" % int(np.ceil(2 * case["l"] / case["c"])), entry.get("n_vertices"), entry.get("max_rel_diff"), flush=True)
json.dump(report, open(os.path.join(ROOT, "tests", "golden", "FUZZ_REPORT.json"), "w"), indent=1)
print("all", len(report), "configurations: oracle == reference")
