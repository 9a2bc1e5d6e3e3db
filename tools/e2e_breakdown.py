#!/usr/bin/env python3
"""tools/e2e_breakdown.py -- where the host-to-host time of one frame goes: pageable input, the reconstruction,
the vertex download, the triangle download (u64 / u32)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context, Parameters
    wl = W.WORKLOADS["s10m_tank"]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                     auto_disable=False, enable_simd=1)
    ctx = Context(0)
    pts = wl["gen"]()
    out = ctx.reconstruct(pts, prm)
    out.mesh_views(u64=True)
    rows = []
    for u64 in (True, False, True, False):
        t0 = time.perf_counter()
        out = ctx.reconstruct(pts, prm, out=out)
        t1 = time.perf_counter()
        v = out.mesh.vertices
        t2 = time.perf_counter()
        t = out.mesh.triangles if u64 else out.mesh.triangles_u32
        t3 = time.perf_counter()
        s = out.stats
        rows.append({"u64": u64, "reconstruct_ms": round((t1 - t0) * 1e3, 3), "ms_upload": round(s["ms_upload"], 3), "ms_total_device": round(s["ms_total"], 3),
                     "vertices_ms": round((t2 - t1) * 1e3, 3), "triangles_ms": round((t3 - t2) * 1e3, 3), "total_ms": round((t3 - t0) * 1e3, 3),
                     "vertex_MB": round(v.nbytes / 1e6, 1), "triangle_MB": round(t.nbytes / 1e6, 1)})
    for r_ in rows:
        print(json.dumps(r_))


if __name__ == "__main__":
    main()
