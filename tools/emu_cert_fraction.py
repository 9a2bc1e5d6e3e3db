"""Certified sub-blocks of a small tank per cube size, on whatever build SPLASHSURF_HIP_LIB names (the CPU execution model of tests/emu when no GPU
is there): python tools/emu_cert_fraction.py [--scale 0.2] [--cubes 0.5,1.0,1.5,2.0] [--simd 0].  Prints one JSON line per cube size with the
certified fraction, the soundness check of tests/test_gpu_certificates.py (smallest complete level-set value inside a certified sub-block) and
the mesh digest, so that two builds can be compared line by line."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--cubes", default="0.5,1.0,1.5,2.0")
    ap.add_argument("--simd", type=int, default=0)
    ap.add_argument("--check", type=int, default=1)
    a = ap.parse_args()
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    pts = W.tank_particles(a.scale)
    from splashsurf_amd.api import Context
    S.load_library()
    ctx2 = Context(0)
    ctx2.set_two_pass(1)
    ctxf = Context(0)
    ctxf.set_full_levelset(True)
    for cube in [float(c) for c in a.cubes.split(",")]:
        kw = dict(particle_radius=0.005, smoothing_length=2.0, cube_size=cube, iso_surface_threshold=0.6, subdomain_grid=True, subdomain_grid_auto_disable=False, simd=bool(a.simd))
        res = S.reconstruct_surface(pts, context=ctx2, **kw)
        masks, bxyz = res.certified_subblocks()
        n_cert = int(np.unpackbits(masks.view(np.uint8)).sum())
        out = {"cube_size": cube, "n_particles": int(pts.shape[0]), "n_active_blocks": int(masks.size), "certified_subblocks": n_cert,
               "certified_frac": round(n_cert / max(1, 8 * masks.size), 4), "n_vertices": int(res.mesh.vertices.shape[0]),
               "mesh_sha": hashlib.sha256(res.mesh.vertices.tobytes() + res.mesh.triangles_u32.tobytes()).hexdigest()[:16],
               "n_large_tile_blocks": int(res.stats.get("n_large_tile_blocks", 0))}
        if a.check:
            full = S.reconstruct_surface(pts, context=ctxf, **kw)
            out["mesh_equals_full_levelset_run"] = bool(np.array_equal(full.mesh.vertices.view(np.uint32), res.mesh.vertices.view(np.uint32)) and np.array_equal(full.mesh.triangles_u32, res.mesh.triangles_u32))
            npnt = [int(x) for x in full.grid.npoints_per_dim]
            G = full.levelset_box([0, 0, 0], npnt)
            worst = np.inf
            n_inside = 0
            for sb in range(8):
                o = bxyz.astype(np.int64) * 8 + np.array([4 * ((sb >> 2) & 1), 4 * ((sb >> 1) & 1), 4 * (sb & 1)], dtype=np.int64)
                inside = np.ones(o.shape[0], dtype=bool)
                exists = o[:, 0] < npnt[0]
                exists &= (o[:, 1] < npnt[1]) & (o[:, 2] < npnt[2])
                mn = np.full(o.shape[0], np.inf)
                for dx in range(4):
                    for dy in range(4):
                        for dz in range(4):
                            x, y, z = np.minimum(o[:, 0] + dx, npnt[0] - 1), np.minimum(o[:, 1] + dy, npnt[1] - 1), np.minimum(o[:, 2] + dz, npnt[2] - 1)
                            mn = np.minimum(mn, G[x, y, z])
                sel = ((masks >> sb) & 1).astype(bool)
                if sel.any():
                    worst = min(worst, float(mn[sel].min()))
                n_inside += int(((mn > 0.6) & exists).sum())
            out["smallest_value_in_a_certified_subblock"] = worst
            out["subblocks_entirely_above_threshold"] = n_inside  # what an exact test would certify
            out["certified_of_certifiable"] = round(n_cert / max(1, n_inside), 4)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
