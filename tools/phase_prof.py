#!/usr/bin/env python3
"""tools/phase_prof.py -- wave-cycles per phase of k_splat_fused from a library built with -DSS_PHASE_PROF
(tools/build_variant.sh prof -- -DSS_PHASE_PROF; SPLASHSURF_HIP_LIB=splashsurf_amd/variants/libsplashsurf_hip_prof.so).
Prints each phase's share of the waves' resident time over one step of the workload."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

NAMES = ["scan", "setup+masks+lists", "classification (all)", "sort", "exact sums", "  cls: outside the walks", "  cls: list walks", "block walk total"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="s10m_tank")
    ap.add_argument("--simd", type=int, default=1)
    a = ap.parse_args()
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context, Parameters, load_library
    lib = load_library()
    wl = W.WORKLOADS[a.workload]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                     auto_disable=False, enable_simd=a.simd)
    ctx = Context(0)
    pts = wl["gen"]()
    out = ctx.reconstruct(pts, prm)
    out = ctx.reconstruct(pts, prm, out=out)
    buf = (C.c_ulonglong * 16)()
    lib.ss_debug_phase_prof(buf, 1)
    out = ctx.reconstruct(pts, prm, out=out)
    lib.ss_debug_phase_prof(buf, 1)
    v = [int(x) for x in buf]
    tot = v[0] + v[7]
    s = out.stats
    print(json.dumps({"workload": a.workload, "simd": a.simd, "ms_levelset_accumulate": s["ms_levelset_accumulate"], "cycles": v[:8]}))
    for i, n in enumerate(NAMES):
        print("%-24s %14d  %5.1f %%" % (n, v[i], 100.0 * v[i] / max(tot, 1)))


if __name__ == "__main__":
    main()
