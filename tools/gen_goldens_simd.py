#!/usr/bin/env python3
"""Goldens of the reference's DEFAULT arithmetic path, `simd=True` (Parameters::enable_simd, lib.rs:179-181).

Build container only: imports the reference's pre-built wheel (tools/oracle_env.py) and runs
`pysplashsurf.reconstruct_surface(..., simd=True, subdomain_grid=True, subdomain_grid_auto_disable=False)` on this
x86-64 host (AVX2+FMA: dense subdomains take density_grid_loop_avx, dense_subdomains.rs:991-1133).

While generating, the oracle's two SIMD modes (oracle/splash_oracle_decl.h: enable_simd) are pinned:
  * mode 1, the lane-by-lane restatement of the AVX loop incl. the unfused remainder lanes and the scalar loop for
    sparse subdomains, must reproduce the wheel's mesh -- same vertices on the same grid edges, same triangle sets,
    coordinates bit-identical except on subdomain faces, where the reference's own result depends on which subdomain's
    patch is stitched first (its AVX values on a shared face differ between the two subdomains);
  * mode 2, that arithmetic applied uniformly to every (particle, point) pair -- what the HIP library computes for
    enable_simd = 1 -- must give the same topology and stay within 1e-5 relative (north_star) of the wheel.
  * densities must not depend on `simd` at all.
Results: tests/golden/simd_*.npz (+ SIMD_REPORT.json with the measured differences).
"""
import hashlib
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
from splashsurf_amd import workloads as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "data")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_run(p, r, l, c, t, n_cubes, simd):
    res = pysplashsurf.reconstruct_surface(p, particle_radius=r, smoothing_length=l, cube_size=c, iso_surface_threshold=t, simd=simd, multi_threading=True,
                                           subdomain_grid=True, subdomain_grid_auto_disable=False, subdomain_num_cubes_per_dim=n_cubes)
    return dict(vertices=np.asarray(res.mesh.vertices, dtype=np.float32).reshape(-1, 3), triangles=np.asarray(res.mesh.triangles).astype(np.int64).reshape(-1, 3),
                densities=np.asarray(res.particle_densities, dtype=np.float32), grid_min=np.asarray(res.grid.aabb.min, dtype=np.float64).astype(np.float32),
                cell_size=np.float32(res.grid.cell_size), n_points=np.asarray(res.grid.npoints_per_dim, dtype=np.int64),
                n_cells=np.asarray(res.grid.ncells_per_dim, dtype=np.int64))


def main():
    cases = [
        # name, input description, r, l, c, t, n_cubes, full mesh stored?
        ("simd_kat1", dict(kind="inline", points=[[0.01, 0.0, 0.0]]), 1.0, 0.5, 1.0, 0.1, 64, True),
        ("simd_cube_2366_n16", dict(kind="file", file="cube_2366_particles.npy"), 0.025, 2.0, 0.75, 0.6, 16, True),
        ("simd_config1_double_dam_break", dict(kind="file", file="double_dam_break_frame_26_4732_particles.npy"), 0.025, 2.0, 1.1, 0.6, 64, True),
        ("simd_config1_n16", dict(kind="file", file="double_dam_break_frame_26_4732_particles.npy"), 0.025, 2.0, 1.1, 0.6, 16, False),
        ("simd_bunny_7705", dict(kind="file", file="bunny_frame_14_7705_particles.npy"), 0.025, 2.0, 0.5, 0.6, 64, False),
        ("simd_config5_hilbert", dict(kind="file", file="hilbert_46843_particles.npy"), 0.025, 2.0, 0.45, 0.6, 64, False),
        ("simd_tank_small", dict(kind="workload", name="tank", scale=0.08), 0.005, 2.0, 0.5, 0.6, 64, False),
    ]
    report = {}
    for name, desc, r, l, c, t, n_cubes, full in cases:
        if desc["kind"] == "inline":
            pts = np.asarray(desc["points"], dtype=np.float32).reshape(-1, 3)
        elif desc["kind"] == "file":
            path = os.path.join(DATA, desc["file"])
            if not os.path.exists(path):
                print("skip", name, "(no input file)")
                continue
            pts = np.load(path)
        else:
            pts = W.tank_particles(scale=desc["scale"])
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        ref = ref_run(pts, r, l, c, t, n_cubes, True)
        ref_scalar = ref_run(pts, r, l, c, t, n_cubes, False)
        assert np.array_equal(ref["densities"].view(np.uint32), ref_scalar["densities"].view(np.uint32)), name + ": the reference's densities depend on simd"
        rec = {}
        for mode in (1, 2):
            orc = O.reconstruct_surface(pts, O.make_params_relative(r, l, c, iso_surface_threshold=t, subdomain_num_cubes_per_dim=n_cubes, simd=mode))
            assert np.array_equal(ref["densities"].view(np.uint32), orc.particle_densities.view(np.uint32)), name
            cmp = MC.compare_geometric(ref["vertices"], ref["triangles"], orc.vertices, orc.triangles, ref["grid_min"], ref["cell_size"], ref["n_points"])
            assert cmp["ids_equal"] and cmp["triangles_equal"], (name, mode, cmp)
            assert cmp["max_rel_diff"] <= (1e-6 if mode == 1 else 1e-5), (name, mode, cmp)
            rec["oracle_mode_%d" % mode] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in cmp.items() if k in ("max_rel_diff", "max_abs_diff", "n_vertices_bit_equal")}
            rec["oracle_mode_%d" % mode]["n_vertices"] = int(ref["vertices"].shape[0])
        # the reference's own two paths against each other (its noise floor; may differ in topology)
        try:
            c01 = MC.compare_geometric(ref["vertices"], ref["triangles"], ref_scalar["vertices"], ref_scalar["triangles"], ref["grid_min"], ref["cell_size"], ref["n_points"])
            rec["reference_simd_vs_scalar"] = dict(ids_equal=bool(c01["ids_equal"]), triangles_equal=bool(c01["triangles_equal"]),
                                                   max_rel_diff=None if not c01["ids_equal"] else float(c01["max_rel_diff"]))
        except AssertionError as e:
            rec["reference_simd_vs_scalar"] = dict(error=str(e)[:100])
        report[name] = rec
        prm = dict(particle_radius=r, smoothing_length=l, cube_size=c, iso_surface_threshold=t, subdomain_num_cubes_per_dim=n_cubes, simd=True)
        common = dict(grid_min=ref["grid_min"], cell_size=ref["cell_size"], n_points=ref["n_points"], n_cells=ref["n_cells"],
                      params=np.array(json.dumps(prm)), input=np.array(json.dumps(desc)), density_sha256=np.array(sha(ref["densities"])))
        if full:
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), vertices=ref["vertices"], triangles=ref["triangles"].astype(np.int32), **common)
        else:
            ids, vs, tc = MC.canonicalize_geometric(ref["vertices"], ref["triangles"], ref["grid_min"], ref["cell_size"], ref["n_points"])
            rng = np.random.default_rng(7)
            sel = np.sort(rng.choice(ids.size, size=min(32768, ids.size), replace=False))
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), n_vertices=np.int64(ids.size), n_triangles=np.int64(tc.shape[0]),
                                ids_sha256=np.array(sha(ids.astype(np.int64))), triangles_sha256=np.array(sha(tc.astype(np.int64))),
                                sample_ids=ids[sel].astype(np.int64), sample_vertices=vs[sel].astype(np.float32), **common)
        print(name, json.dumps(rec))
    json.dump(report, open(os.path.join(GOLD, "SIMD_REPORT.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
