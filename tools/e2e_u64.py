#!/usr/bin/env python3
"""tools/e2e_u64.py [workload] -- host-to-host frames with u64 triangle indices (ss_result_triangles: chunked u32 download, widening
on the context's host threads): ms per frame (best of 5) and a check of the widened indices against the u32 ones."""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "s10m_tank"
    wl = W.WORKLOADS[name]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                     auto_disable=False, enable_simd=1)
    ctx = Context(0)
    pts = wl["gen"]()
    out = ctx.reconstruct(pts, prm)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = ctx.reconstruct(pts, prm, out=out)
        v, t = out.mesh_views(u64=True)
        ts.append(time.perf_counter() - t0)
    t32 = out.mesh.triangles_u32
    ok = bool(t.dtype == np.uint64 and t.shape == t32.shape and np.array_equal(t, t32.astype(np.uint64)))
    print(json.dumps({"workload": name, "n": int(pts.shape[0]), "n_triangles": int(t.shape[0]), "ms_per_frame_best": round(min(ts) * 1e3, 3),
                      "mparticles_per_s": round(pts.shape[0] / min(ts) / 1e6, 1), "u64_equals_u32": ok}))
    ctx.close()


if __name__ == "__main__":
    main()
