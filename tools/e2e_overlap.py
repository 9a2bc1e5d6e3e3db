#!/usr/bin/env python3
"""tools/e2e_overlap.py -- do two frames in flight (two contexts / streams / host threads) overlap their PCIe copies with each
other's kernels?  ms per frame for input in {pageable host, HBM} x output in {left in HBM, host views}, one and two threads."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context, Parameters
    wl = W.WORKLOADS["s10m_tank"]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                     auto_disable=False, enable_simd=1)
    pts = wl["gen"]()
    dev = torch.from_numpy(pts).to("cuda:0")
    pinned = torch.from_numpy(pts).pin_memory()
    torch.cuda.synchronize()
    ctxs = [Context(0), Context(0)]
    outs = [None, None]
    for i in range(2):
        outs[i] = ctxs[i].reconstruct(pts, prm)
        outs[i].mesh_views()
    frames = 4
    for inp_name, inp in (("pageable", pts), ("pinned", pinned), ("hbm", dev)):
        for down in (False, True):
            for nthreads in (1, 2):
                def worker(i):
                    for _ in range(frames):
                        outs[i] = ctxs[i].reconstruct(inp, prm, out=outs[i])
                        if down:
                            outs[i].mesh_views()
                th = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(json.dumps({"input": inp_name, "download": down, "threads": nthreads, "ms_per_frame": round(dt / (frames * nthreads) * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
