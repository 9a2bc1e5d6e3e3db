"""Builds tests/emu/_build/libsplashsurf_emu.so: the library's own sources (splashsurf_amd/csrc/*.hip, unmodified) compiled as C++ for
the host against tests/emu/include/hip/hip_runtime.h -- the CPU execution model of the kernels.  TEST INFRASTRUCTURE: only tests load it.

    python tests/emu/build_emu.py [--force] [--asan] [-DNAME=VALUE ...] [--out PATH]

--asan: the same with -fsanitize=address (libsplashsurf_emu_asan.so): AddressSanitizer over the KERNELS' loads and stores -- the GPU pool offers no device
sanitizer --; run with LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 (tests/emu/run_asan.sh).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "splashsurf_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libsplashsurf_emu.so")
SOURCES = ["ss_api.hip", "ss_kernels.hip", "ss_global.hip", "ss_post.hip", "ss_dist.hip", "ss_prims.hip", "ss_pipeline.hip"]
CXX_CANDIDATES = ["/opt/rocm/lib/llvm/bin/clang++", "clang++"]
# the device build's floating-point contract (no contraction; fma only where the source says fma) on the host's IEEE arithmetic
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mf16c", "-mavx2", "-fno-slp-vectorize",
         "-fno-vectorize", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-variable", "-Wno-unknown-pragmas",
         "-Wno-pass-failed", "-Wno-ignored-attributes", "-Wno-unused-but-set-variable", "-Wno-unused-local-typedef",
         "-I", os.path.join(HERE, "include")]


def cxx():
    for c in CXX_CANDIDATES:
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return CXX_CANDIDATES[-1]


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, defines=(), out=OUT):
    objdir = os.path.join(os.path.dirname(out), "obj_" + os.path.basename(out))
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers += [os.path.join(ROOT, "include", "splashsurf_hip.h"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                os.path.join(HERE, "include", "rccl", "rccl.h"), os.path.abspath(__file__)]
    units = [(os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))) for s in SOURCES]
    units.append((os.path.join(HERE, "emu_runtime.cpp"), os.path.join(objdir, "emu_runtime.o")))

    def compile_one(unit):
        src, obj = unit
        if force or stale(obj, [src] + headers):
            cmd = [cxx()] + FLAGS + list(defines) + ["-c", src, "-o", obj]
            print("[emu-build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, units))
    if force or stale(out, objs):
        cmd = [cxx(), "-shared", "-fPIC", "-Wl,-Bsymbolic"] + objs  # (-Bsymbolic: the hip* calls bind to the emulator even when a real libamdhip64 is loaded in the process)
        cmd += [d for d in defines if d.startswith("-fsanitize")] + ["-ldl", "-lpthread", "-o", out]
        print("[emu-build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_fake_rccl(force=False):
    """tests/emu/_build/libfake_rccl.so: the stand-in RCCL of the multi-process tests (fake_rccl.cpp), bound through SPLASH_RCCL_LIB."""
    src = os.path.join(HERE, "fake_rccl.cpp")
    out = os.path.join(HERE, "_build", "libfake_rccl.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or stale(out, [src]):
        # (-Bsymbolic: its own nccl* calls stay inside it even when a real librccl is loaded in the process)
        cmd = [cxx(), "-std=c++17", "-O1", "-g1", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-Wall", src, "-o", out]
        print("[emu-build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    if "--fake-rccl" in args:
        print(build_fake_rccl(force="--force" in args))
        sys.exit(0)
    out = OUT
    if "--out" in args:
        i = args.index("--out")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    extra = [a for a in args if a.startswith("-D")]
    if "--asan" in args:
        extra += ["-fsanitize=address", "-fno-omit-frame-pointer", "-O1"]
        if out == OUT:
            out = OUT.replace("_emu.so", "_emu_asan.so")
    print(build(force="--force" in args, defines=extra, out=out))
