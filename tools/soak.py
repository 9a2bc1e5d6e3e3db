#!/usr/bin/env python3
"""tools/soak.py -- run-to-run determinism at benchmark scale: N frames of S10M-tank on one context, every frame's densities / vertices /
triangles hashed; all digests must agree (a race or an order-dependent sum would show as a differing frame)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from splashsurf_amd import workloads as W
from splashsurf_amd.api import Context, Parameters
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
out_lines = []
for simd in (0, 1):
    wl = W.WORKLOADS["s10m_tank"]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r), auto_disable=False, enable_simd=simd)
    ctx = Context(0)
    d = torch.from_numpy(wl["gen"]()).to("cuda:0")
    res, digests = None, []
    for f in range(n_frames):
        res = ctx.reconstruct(d, prm, out=res)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(res.particle_densities).tobytes())
        v, t = res.mesh_views(u64=False)
        h.update(np.ascontiguousarray(v).tobytes())
        h.update(np.ascontiguousarray(t).tobytes())
        digests.append(h.hexdigest()[:16])
    line = {"workload": "s10m_tank", "enable_simd": simd, "frames": n_frames, "distinct_digests": sorted(set(digests)), "n_vertices": int(res.stats["n_vertices"]),
            "n_triangles": int(res.stats["n_triangles"]), "identical": len(set(digests)) == 1}
    print(json.dumps(line), flush=True)
    ctx.close()
