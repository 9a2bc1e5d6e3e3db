"""tests/emu/bench_dry_run.py [bench.py arguments]: bench.py's OWN code on the CPU execution model of the kernels (test infrastructure; SPLASHSURF_HIP_LIB must
name tests/emu/_build/libsplashsurf_emu.so).

bench.py refuses to run without a GPU, and nothing in it knows about the emulator.  This wrapper runs it unchanged with torch's CUDA entry points pointed at
the host (the emulated library's "device" memory is host memory), the workload generators shrunk to a few thousand particles and the HBM probe to 16 MiB, so
that every line of the default record -- the timed loop, the arithmetic modes, host-to-host frames, the frame pipeline, the post-processing recipe, the other
configurations, the pseudo-rank mode -- executes in the no-GPU container.  What the JSON line then holds is meaningless as a measurement (and says so:
`"dry_run": true` is added to it); the test (tests/test_emu_kernels.py) checks that the record is complete and that no extra failed."""
import json
import os
import runpy
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)

if not os.environ.get("SPLASHSURF_HIP_LIB"):
    raise SystemExit("SPLASHSURF_HIP_LIB must name the emulated library")

import numpy as np  # noqa: E402
import torch  # noqa: E402

_device = torch.device
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *a, **k: None
torch.cuda.synchronize = lambda *a, **k: None


class _HostDevice:
    def __call__(self, *a, **k):
        return _device("cpu")


torch.device = _HostDevice()

import torch.distributed as dist  # noqa: E402

_init = dist.init_process_group


def _init_on_host(backend=None, device_id=None, **kw):  # (N > 1 under torch.distributed.run: bench.py asks for "nccl"; the ranks here are CPU processes)
    return _init(backend="gloo", **kw)


dist.init_process_group = _init_on_host

from splashsurf_amd import api, workloads as W  # noqa: E402

# the benchmark's workloads, a few thousand particles each (same shapes: lattice tank, uniform-random cube)
W.WORKLOADS["s10m_tank"]["gen"] = lambda: W.tank_particles(scale=0.08)
W.WORKLOADS["s40m_tank"]["gen"] = lambda: W.tank_particles(scale=0.1)
W.WORKLOADS["s1m"]["gen"] = lambda: W.uniform_cube_particles(3000, seed=1) * np.float32(0.25)
W.WORKLOADS["s10m_cube"]["gen"] = lambda: W.uniform_cube_particles(6000, seed=2) * np.float32(0.12)
_measure = api.Context.measure_hbm_bandwidth
api.Context.measure_hbm_bandwidth = lambda self, nbytes=0, repetitions=1: _measure(self, 16 << 20, 1)

_print = print
lines = []


def _capture(*a, **k):
    if len(a) == 1 and isinstance(a[0], str) and a[0].startswith("{\"metric\""):
        d = json.loads(a[0])
        d["dry_run"] = True
        lines.append(d)
        return _print(json.dumps(d), **k)
    return _print(*a, **k)


import builtins  # noqa: E402

builtins.print = _capture
script = os.path.join(ROOT, "bench.py")
if len(sys.argv) > 2 and sys.argv[1] == "--script":  # (the measurement tools of tools/ take the same treatment: tools/final_collect.sh's steps can be rehearsed)
    script = os.path.abspath(sys.argv[2])
    del sys.argv[1:3]
sys.argv = [script] + sys.argv[1:]
runpy.run_path(script, run_name="__main__")
