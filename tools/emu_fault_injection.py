#!/usr/bin/env python3
"""tools/emu_fault_injection.py -- allocation failures at every allocation point of a reconstruction, on the CPU execution model of tests/emu (the
emulated hipMalloc / hipHostMalloc can be told to fail their n-th call; no real device can be made to do that on demand).

    SPLASHSURF_HIP_LIB=tests/emu/_build/libsplashsurf_emu.so python tools/emu_fault_injection.py [--out profiles/r06_emu_fault_injection.jsonl]

For k = 1, 2, ...: a FRESH context, the k-th device (then: pinned-host) allocation of its first call fails.  Expected of the library: the call returns an
error (SplashsurfError with the HIP error text: SS_ERR_DEVICE) instead of crashing or handing back a mesh, and the SAME context then completes the same
call with the injection off and produces the reference digest -- a failed grow-only buffer must not leave a dangling pointer or a stale capacity behind.
Stops at the first k that no longer reaches an allocation.  Covers the subdomain-grid path (with the over-dense branch), the global strategy, f64 and
the post-processing entry points."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(res):
    h = hashlib.sha256()
    for a in (res.mesh.vertices, res.mesh.triangles_u32, res.particle_densities):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-k", type=int, default=400)
    a = ap.parse_args()
    lib_path = os.environ.get("SPLASHSURF_HIP_LIB", "")
    if "emu" not in os.path.basename(lib_path):
        sys.exit("SPLASHSURF_HIP_LIB must name the emulated library (tests/emu/build_emu.py): only its allocator can be told to fail")
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context, SplashsurfError
    L = S.load_library()
    emu = C.CDLL(lib_path)
    emu.hip_emu_fail_malloc_in.argtypes = [C.c_int, C.c_longlong]
    emu.hip_emu_malloc_count.argtypes = [C.c_int]
    emu.hip_emu_malloc_count.restype = C.c_ulonglong
    tank = W.tank_particles(0.08)
    dense = (np.random.default_rng(5).random((20000, 3)) * 0.2).astype(np.float32)
    scenarios = {
        "grid_f32": lambda ctx: S.reconstruct_surface(tank, context=ctx, particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, simd=False),
        "grid_f32_over_dense": lambda ctx: S.reconstruct_surface(dense, context=ctx, particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, simd=True),
        "grid_f64": lambda ctx: S.reconstruct_surface(tank.astype(np.float64), context=ctx, particle_radius=0.005, smoothing_length=2.0, cube_size=0.75, subdomain_grid_auto_disable=False),
        "global_f32": lambda ctx: S.reconstruct_surface(tank[::4].copy(), context=ctx, particle_radius=0.005, smoothing_length=2.0, cube_size=1.0, subdomain_grid=False),
    }
    out = open(a.out, "w") if a.out else None
    summary = {}
    bad = 0
    for name, run in scenarios.items():
        ref = digest(run(Context(0)))
        for host in (0, 1):
            k, reached = 1, 0
            while k <= a.max_k:
                ctx = Context(0)
                before = emu.hip_emu_malloc_count(host)
                emu.hip_emu_fail_malloc_in(host, k)
                rec = dict(scenario=name, kind="pinned" if host else "device", k=k)
                try:
                    res = run(ctx)
                    rec["first"] = "completed"
                    rec["digest_ok"] = digest(res) == ref
                except SplashsurfError as e:
                    rec["first"] = "error"
                    rec["message"] = str(e)[:160]
                emu.hip_emu_fail_malloc_in(host, 0)
                hit = emu.hip_emu_malloc_count(host) - before >= k
                rec["injection_reached"] = bool(hit)
                try:
                    rec["second_digest_ok"] = digest(run(ctx)) == ref
                except SplashsurfError as e:
                    rec["second_digest_ok"] = False
                    rec["second_message"] = str(e)[:160]
                ctx.close()
                ok = rec["second_digest_ok"] and ((rec["first"] == "error") if hit else (rec["first"] == "completed" and rec["digest_ok"]))
                rec["ok"] = bool(ok)
                bad += not ok
                if out:
                    out.write(json.dumps(rec) + "\n")
                if not ok:
                    print(json.dumps(rec), flush=True)
                if not hit:
                    break
                reached += 1
                k += 1
            summary["%s/%s" % (name, "pinned" if host else "device")] = reached
    line = json.dumps(dict(summary=True, allocation_points_failed_one_by_one=summary, failures=bad))
    print(line)
    if out:
        out.write(line + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
