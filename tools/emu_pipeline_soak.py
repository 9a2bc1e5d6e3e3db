"""tools/emu_pipeline_soak.py: the frame pipeline (ss_pipeline_*) under a random series -- 80 frames of three clouds cut at random, 15 % of them with a failing parameter set, depths 1-4, random
pauses of the consumer -- on the library SPLASHSURF_HIP_LIB names (the CPU execution model of tests/emu, or the HIP build on a GPU box): every good frame equals the direct call bit for
bit, every bad frame raises at its own next().  Round 6: 71 + 9 frames, 22 s emulated."""
import sys, numpy as np, random, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from splashsurf_amd.api import FramePipeline, Parameters, Context, SplashsurfError
from conftest import golden_input, load_golden, load_points
rng = random.Random(7)
bases = [load_points("double_dam_break_frame_26_4732_particles.npy").astype(np.float32), golden_input(load_golden("cube_2366")).astype(np.float32), golden_input(load_golden("bunny_7705")).astype(np.float32)]
good = Parameters.new_relative(0.025, 4.0, 0.9, auto_disable=False, enable_simd=False)
bad = Parameters.new_relative(0.025, 4.0, 0.9, auto_disable=False); bad.cube_size = 0.0
ctx = Context(0)
def key(res):
    v, t = res.mesh_views(); return (v.tobytes(), t.tobytes())
cache = {}
def direct(pts):
    k = pts.tobytes()
    if k not in cache:
        cache[k] = key(ctx.reconstruct(pts, good))
    return cache[k]
n_ok = n_fail = 0
t0 = time.time()
for depth in (1, 2, 3, 4):
    with FramePipeline(0, depth) as pipe:
        frames = []
        for k in range(20):
            b = bases[rng.randrange(3)]
            cut = rng.randrange(0, len(b) // 3)
            pts = np.ascontiguousarray(b[cut:] + np.float32(0.001 * rng.randrange(10)))
            frames.append((pts, rng.random() < 0.15))
        it = iter(frames); inflight = []
        def feed():
            while pipe.in_flight < depth:
                try: pts, isbad = next(it)
                except StopIteration: return
                pipe.submit(pts, bad if isbad else good); inflight.append((pts, isbad))
        feed()
        while inflight:
            pts, isbad = inflight.pop(0)
            if rng.random() < 0.3: time.sleep(0.02)
            try:
                t, r = pipe.next()
                assert not isbad
                assert key(r) == direct(pts), "frame differs"
                n_ok += 1
            except SplashsurfError as e:
                assert isbad and e.status == 4, (isbad, e)
                n_fail += 1
            feed()
print("pipeline soak: %d frames equal to the direct call, %d failing frames reported, depths 1-4, %.0f s" % (n_ok, n_fail, time.time() - t0))
