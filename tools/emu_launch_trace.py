"""tools/emu_launch_trace.py [config1|config5]: what ONE steady-state reconstruct call dispatches, read from the launch trace of the CPU execution model
(tests/emu, HIP_EMU_TRACE=1; SPLASHSURF_HIP_LIB must name the emulated library).  Prints one JSON object: the kernels in order, the memsets with their sizes,
the copies, ss_stats.n_host_waits.  A small job's time IS this list (config 1 = ~45 dispatches of 4-70 us and 8 host waits), and the
list does not depend on the device -- the host flow is the same code on the GPU -- so the counts are checked in the CPU suite
(tests/test_emu_kernels.py::test_launch_trace_of_a_small_job)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CASES = {"config1": ("double_dam_break_frame_26_4732_particles.npy", 1.1), "config5": ("hilbert_46843_particles.npy", 0.45)}

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from splashsurf_amd.api import Parameters, Context
pts = np.load(%(data)r)
prm = Parameters(particle_radius=0.025, compact_support_radius=2.0 * 2.0 * 0.025, cube_size=%(cube)r * 0.025, enable_simd=False)  # (f64 products, like pysplashsurf: reconstruction.rs:171-193)
ctx = Context(0)
out = ctx.reconstruct(pts, prm)
out = ctx.reconstruct(pts, prm, out=out)           # (buffers sized, row table made, division verified)
print("=====BEGIN", file=sys.stderr, flush=True)
out = ctx.reconstruct(pts, prm, out=out)
print("=====END", file=sys.stderr, flush=True)
print("WAITS", out.stats["n_host_waits"], *out.counts())
"""


def trace(case):
    data, cube = CASES[case]
    lib = os.environ.get("SPLASHSURF_HIP_LIB")
    if not lib:
        raise SystemExit("SPLASHSURF_HIP_LIB must name tests/emu/_build/libsplashsurf_emu.so")
    env = dict(os.environ, HIP_EMU_TRACE="1")
    code = CHILD % {"root": ROOT, "data": os.path.join(ROOT, "tests", "data", data), "cube": cube}
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    inside, kernels, memsets, copies = False, [], [], []
    for line in p.stderr.splitlines():
        if line.startswith("=====BEGIN"):
            inside = True
        elif line.startswith("=====END"):
            inside = False
        elif inside and line.startswith("[hip-emu]"):
            m = re.match(r"\[hip-emu\] launch \(?([A-Za-z_0-9]+)", line)
            if m:
                kernels.append(m.group(1))
                continue
            m = re.match(r"\[hip-emu\] memsetAsync (\d+) bytes", line)
            if m:
                memsets.append(int(m.group(1)))
                continue
            m = re.match(r"\[hip-emu\] memcpy\w* (\d+) bytes", line)
            if m:
                copies.append(int(m.group(1)))
    w = [l for l in p.stdout.splitlines() if l.startswith("WAITS")][0].split()
    return {"case": case, "launches": len(kernels), "memsets": memsets, "copies": copies, "n_host_waits": int(w[1]), "n_vertices": int(w[2]), "n_triangles": int(w[3]),
            "kernels": kernels}


if __name__ == "__main__":
    print(json.dumps(trace(sys.argv[1] if len(sys.argv) > 1 else "config1")))
