"""tools/emu_pipeline_faults.py: allocation failures inside a frame of the frame pipeline (ss_pipeline_*), on the CPU execution model of tests/emu -- the k-th device (then: pinned-host)
allocation of the first frame fails on its slot's thread, for every k that is reached: the frame must report an error at next(), and two more frames through the same pipeline must give the
reference digest.  The summary line is appended to profiles/r06_emu_fault_injection.jsonl."""
import ctypes as C, os, sys, hashlib, json
import numpy as np
sys.path.insert(0, '/root/repo')
import splashsurf_amd as S
from splashsurf_amd import workloads as W
from splashsurf_amd.api import Context, FramePipeline, Parameters, SplashsurfError
lib_path = os.environ["SPLASHSURF_HIP_LIB"]
S.load_library()
emu = C.CDLL(lib_path)
emu.hip_emu_fail_malloc_in.argtypes = [C.c_int, C.c_longlong]
emu.hip_emu_malloc_count.argtypes = [C.c_int]
emu.hip_emu_malloc_count.restype = C.c_ulonglong
tank = W.tank_particles(0.08)
prm = Parameters(particle_radius=0.005, compact_support_radius=0.02, cube_size=0.0025, enable_simd=False, auto_disable=False)
def digest(res):
    v, t = res.mesh_views()
    return hashlib.sha256(v.tobytes() + t.tobytes() + res.particle_densities.tobytes()).hexdigest()[:16]
ref = digest(Context(0).reconstruct(tank, prm))
bad = 0; summary = {}
for host in (0, 1):
    k = 1; reached = 0
    while k <= 300:
        pipe = FramePipeline(0, 2)
        before = emu.hip_emu_malloc_count(host)
        emu.hip_emu_fail_malloc_in(host, k)
        pipe.submit(tank, prm, FramePipeline.FETCH_VERTICES | FramePipeline.FETCH_TRIANGLES_U32 | FramePipeline.FETCH_DENSITIES)
        first = None
        try:
            _, r = pipe.next(); first = "completed" if digest(r) == ref else "WRONG"
        except SplashsurfError as e:
            first = "error"
        emu.hip_emu_fail_malloc_in(host, 0)
        hit = emu.hip_emu_malloc_count(host) - before >= k
        # the same slot again (frame 2 runs on slot 0 again with depth 2 only after frame 1 on slot 1): two more frames
        ok2 = True
        for _ in range(2):
            pipe.submit(tank, prm)
        for _ in range(2):
            try:
                _, r = pipe.next(); ok2 = ok2 and digest(r) == ref
            except SplashsurfError as e:
                ok2 = False
        pipe.close()
        ok = ok2 and (first == "error" if hit else first == "completed")
        if not ok:
            bad += 1; print(json.dumps(dict(kind=host, k=k, first=first, hit=bool(hit), ok2=ok2)), flush=True)
        if not hit: break
        reached += 1; k += 1
    summary["pinned" if host else "device"] = reached
print(json.dumps(dict(summary=True, scenario="frame pipeline (depth 2): the k-th allocation of a frame fails on its slot's thread; the frame reports the error at next(), the pipeline and the slot go on with the right digest", allocation_points_failed_one_by_one=summary, failures=bad)))
