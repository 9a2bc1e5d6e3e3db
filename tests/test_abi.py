"""CPU tests: the C-ABI library loads and exports every symbol include/splashsurf_hip.h declares;
the Python host mirrors pysplashsurf.reconstruct_surface.  No compute calls (no GPU here)."""
import ctypes
import inspect
import os

import pytest

import __graft_entry__ as G


def test_library_exports_all_declared_symbols():
    lib = G.LIB
    if not os.path.exists(lib):
        G.build()
    L = ctypes.CDLL(lib)
    syms = G.exported_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), s
    L.ss_abi_version.restype = ctypes.c_int
    assert L.ss_abi_version() == 6  # SS_ABI_VERSION in include/splashsurf_hip.h


def test_struct_layouts_match_header():
    from splashsurf_amd import api
    assert ctypes.sizeof(api._Params) == 5 * 4 + 4 + 24 + 6 * 4
    assert ctypes.sizeof(api._Grid) == 28 + 4 + 48  # 7 floats + pad + 6 int64
    # + gather/accumulate timings, large-tile block count (ABI 2), arithmetic mode and arena bytes used / reserved (ABI 3), MC block count and the
    # density / MC kernel timers (ABI 4)
    assert ctypes.sizeof(api._Stats) == 9 * 8 + 8 * 8 + 3 * 8 + 7 * 8 + 4 * 8 + 8


def test_python_signature_mirrors_reference():
    import splashsurf_amd as S
    sig = inspect.signature(S.reconstruct_surface)
    names = list(sig.parameters)
    # pysplashsurf/src/reconstruction.rs:135-160
    for n in ["particles", "particle_radius", "rest_density", "smoothing_length", "cube_size", "iso_surface_threshold", "aabb_min",
              "aabb_max", "multi_threading", "simd", "global_neighborhood_list", "subdomain_grid", "subdomain_grid_auto_disable",
              "subdomain_num_cubes_per_dim"]:
        assert n in names, n
    assert sig.parameters["rest_density"].default == 1000.0
    assert sig.parameters["iso_surface_threshold"].default == 0.6
    assert sig.parameters["subdomain_num_cubes_per_dim"].default == 64


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly instead of computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from splashsurf_amd.api import Context, SplashsurfError
    with pytest.raises(SplashsurfError):
        Context(0)


def test_cpp_header_compiles():
    """include/splashsurf_hip.hpp (the C++ mirror of the reference's Rust API, incl. the multi-GPU wrapper) and its test
    program are valid C++17 for a plain host compiler."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for src in ("test_host.cpp",):
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", src)])


def test_dist_info_layout_matches_the_c_header(tmp_path):
    """ss_dist_info as the C compiler lays it out (gcc on include/splashsurf_hip.h) == the ctypes mirror the Python host reads it through
    (splashsurf_amd/distributed.py): size and the offset of every field (round 6 appended bytes_link_max)."""
    import shutil
    import subprocess
    from splashsurf_amd import distributed as D
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    fields = [n for n, _ in D._DistInfo._fields_]
    root = os.path.join(os.path.dirname(__file__), "..")
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "splashsurf_hip.h"\nint main(void) {\n  printf("%zu\\n", sizeof(ss_dist_info));\n' +
                   "".join('  printf("%%zu\\n", offsetof(ss_dist_info, %s));\n' % f for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(D._DistInfo)
    for name, off in zip(fields, out[1:]):
        assert getattr(D._DistInfo, name).offset == off, name
