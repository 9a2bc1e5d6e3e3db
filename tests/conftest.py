import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "tests", "data")
GOLD = os.path.join(ROOT, "tests", "golden")


if os.environ.get("SPLASHSURF_EMU_HOST_DEVICE") == "1":
    # tests/test_emu_kernels.py only: the library under test is the CPU execution model of tests/emu, whose "device" memory is the host's, so host code that
    # asks torch for a "cuda" device (postprocessing.reconstruction_pipeline, the CLI) gets the host.  Never set on a GPU box.
    import torch

    _torch_device = torch.device

    class _HostDevice:
        def __call__(self, *a, **k):
            return _torch_device("cpu")

    torch.device = _HostDevice()
    torch.cuda.synchronize = lambda *a, **k: None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def device_name():
    """"cuda:0" on a GPU box; "cpu" when the library under test is the CPU execution model of tests/emu (SPLASHSURF_HIP_LIB), whose "device" memory is the host's:
    tests that hand the library torch tensors then hand it host tensors -- the arithmetic under test is the same, the HBM-resident input path is not exercised."""
    import torch
    return "cuda:0" if torch.cuda.is_available() else "cpu"


def device_sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def load_points(name):
    return np.load(os.path.join(DATA, name))


def load_golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def golden_input(g):
    """Re-create the input particles of a golden fixture from its description."""
    import json
    from splashsurf_amd import workloads as W
    d = json.loads(str(g["input"]))
    if d["kind"] == "inline":
        return np.asarray(d["points"], dtype=np.float32).reshape(-1, 3)
    if d["kind"] == "file":
        return load_points(d["file"])
    if d["kind"] == "workload":
        if d["name"] == "tank":
            return W.tank_particles(scale=d["scale"])
        if d["name"] == "uniform_cube":
            return W.uniform_cube_particles(d["n"], seed=d["seed"])
    raise KeyError(d)


def golden_params(g):
    import json
    return json.loads(str(g["params"]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    """HIP context on GPU 0. Fails loudly (no CPU fallback) if the library or the GPU is missing."""
    import splashsurf_amd as S
    from splashsurf_amd.api import Context
    S.load_library()
    return Context(0)


@pytest.fixture(scope="session")
def full_levelset_ctx():
    """A second context with SS_OPTION_FULL_LEVELSET: no early exit inside the fluid, every level-set value complete --
    for the tests that compare whole 65^3 level-set arrays (the default context only completes what marching cubes reads)."""
    import splashsurf_amd as S
    from splashsurf_amd.api import Context
    S.load_library()
    ctx = Context(0)
    ctx.set_full_levelset(True)
    return ctx


@pytest.fixture(scope="session")
def two_pass_ctx():
    """A context with SS_OPTION_SPLAT_TWO_PASS = 1: the splat certifies sub-blocks inside the fluid and completes only what
    marching cubes reads, also on small jobs (the automatic setting reserves the scheme for >= 1 k active blocks and switches it off per workload)."""
    import splashsurf_amd as S
    from splashsurf_amd.api import Context
    S.load_library()
    ctx = Context(0)
    ctx.set_two_pass(1)
    return ctx
