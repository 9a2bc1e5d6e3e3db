#!/usr/bin/env python3
"""First-contact diagnostics on the GPU box: run the HIP path on a few inputs and PRINT how it
compares with the oracle instead of asserting (one gpurun call => as much information as possible)."""
import os, sys, time, traceback
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import splashsurf_amd as S
from splashsurf_amd import workloads as W
from splashsurf_amd.api import Context
from oracle import oracle as O
import mesh_compare as MC

ctx = Context(0)

def case(name, pts, r, l, c, t=0.6, n_cubes=64, levelset=True):
    print("=== %s: n=%d r=%g l=%g c=%g" % (name, pts.shape[0], r, l, c), flush=True)
    try:
        t0 = time.time()
        res = S.reconstruct_surface(pts, particle_radius=r, smoothing_length=l, cube_size=c, iso_surface_threshold=t,
                                    subdomain_grid_auto_disable=False, subdomain_num_cubes_per_dim=n_cubes, context=ctx)
        t1 = time.time()
        st = res.stats
        print("  gpu: %.1f ms wall; stats: %s" % ((t1 - t0) * 1e3, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
        par = O.make_params_relative(r, l, c, iso_surface_threshold=t, subdomain_num_cubes_per_dim=n_cubes)
        orc = O.reconstruct_surface(pts, par)
        print("  oracle: %.2f s; v=%d t=%d" % (orc.timings["total"], orc.vertices.shape[0], orc.triangles.shape[0]))
        print("  grid gpu", res.grid.ncells_per_dim, res.grid.aabb.min, "oracle", orc.grid["n_cells"], orc.grid["aabb_min"])
        rho = res.particle_densities
        bad = rho.view(np.uint32) != orc.particle_densities.view(np.uint32)
        rel = np.abs(rho - orc.particle_densities) / np.maximum(orc.particle_densities, 1e-30)
        print("  rho: %d / %d differ; max rel %.3g" % (int(bad.sum()), rho.size, float(rel.max()) if rho.size else 0.0))
        v, k, tr = res.mesh.vertices, res.vertex_keys, res.mesh.triangles
        print("  mesh gpu v=%d t=%d" % (v.shape[0], tr.shape[0]))
        cmp = MC.compare_keyed(v, k, tr, orc.vertices, orc.vertex_keys, orc.triangles)
        print("  keyed compare:", cmp)
        if levelset and pts.shape[0] > 0:
            ns = res.subdomain_grid.ncells_per_dim
            nchk = 0
            for flat in range(ns[0] * ns[1] * ns[2]):
                cnt, ref = O.levelset_subdomain(pts, par, flat)
                if cnt < 0:
                    continue
                s = (flat // (ns[1] * ns[2]), (flat // ns[2]) % ns[1], flat % ns[2])
                got = res.levelset_box([s[0] * n_cubes, s[1] * n_cubes, s[2] * n_cubes], [n_cubes + 1] * 3)
                nb = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
                print("  levelset subdomain %d (%d particles): %d values differ, max abs %.3g, nonzero ref %d gpu %d" %
                      (flat, cnt, nb, float(np.abs(got - ref).max()), int((ref != 0).sum()), int((got != 0).sum())))
                nchk += 1
                if nchk >= 4:
                    break
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()

D = os.path.join(ROOT, "tests", "data")
case("kat1", np.array([[0.01, 0, 0]], np.float32), 1.0, 0.5, 1.0, t=0.1)
case("single", np.array([[0.3, 0.2, 0.1]], np.float32), 0.025, 2.0, 1.0)
case("empty", np.zeros((0, 3), np.float32), 0.025, 2.0, 1.0)
case("cube_8", np.load(os.path.join(D, "cube_8_particles.npy")), 0.025, 2.0, 1.0)
case("cube_2366_n16", np.load(os.path.join(D, "cube_2366_particles.npy")), 0.025, 2.0, 0.75, n_cubes=16)
case("config1", np.load(os.path.join(D, "double_dam_break_frame_26_4732_particles.npy")), 0.025, 2.0, 1.1)
case("free125", np.load(os.path.join(D, "free_particles_125_particles.npy")), 0.025, 2.0, 1.0)
case("tank_small", W.tank_particles(0.08), 0.005, 2.0, 0.5, levelset=False)
case("hilbert", np.load(os.path.join(D, "hilbert_46843_particles.npy")), 0.025, 2.0, 0.45, levelset=False)
case("dense60k", (W.uniform_cube_particles(60000, 99) * np.float32(0.25)).astype(np.float32), 0.01, 2.0, 1.0, levelset=False)
case("s1m", W.uniform_cube_particles(1_000_000, 12345), 0.01, 2.0, 1.0, levelset=False)
