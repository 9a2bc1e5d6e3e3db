#!/usr/bin/env python3
"""tools/e2e_frames.py -- host-to-host frames of S10M-tank (pageable numpy input -> vertices + u64 indices in pinned host memory):
median / best of N frames and the split reconstruct / accessors.  (Round 4 measured an own staged upload with this tool -- host threads copying 4-MB
chunks into pinned slots while earlier chunks are on the link, instead of one hipMemcpyAsync from pageable memory: 13.1-13.3 ms per reconstruct call
against 12.85 ms, i.e. the runtime's own pageable path is the faster one; dropped.)"""
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
    pts = wl["gen"]()
    for rep in range(2):
        ctx = Context(0)
        out = ctx.reconstruct(pts, prm)
        out.mesh_views(u64=True)
        rec, acc, tot = [], [], []
        for _ in range(12):
            t0 = time.perf_counter()
            out = ctx.reconstruct(pts, prm, out=out)
            t1 = time.perf_counter()
            out.mesh_views(u64=True)
            t2 = time.perf_counter()
            rec.append(t1 - t0)
            acc.append(t2 - t1)
            tot.append(t2 - t0)
        med = lambda v: sorted(v)[len(v) // 2] * 1e3
        print(json.dumps({"run": rep, "frame_ms_median": round(med(tot), 3), "frame_ms_best": round(min(tot) * 1e3, 3), "reconstruct_ms_median": round(med(rec), 3),
                          "accessors_ms_median": round(med(acc), 3), "device_ms_total": round(out.stats["ms_total"], 3), "ms_upload_event": round(out.stats["ms_upload"], 3),
                          "Mparticles_per_s_median": round(pts.shape[0] / med(tot) / 1e3, 1)}))
        out._free()
        ctx.close()


if __name__ == "__main__":
    main()
