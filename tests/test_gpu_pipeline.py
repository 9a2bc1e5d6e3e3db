"""ss_pipeline_* (include/splashsurf_hip.h, csrc/ss_pipeline.hip): a time series of frames through `depth` contexts of one device.

The reference's counterpart is the loop of reconstruct_surface_inplace over the frames of a series with one workspace (lib.rs:340-346, 466-470): every
frame's output must be exactly what that loop gives -- here: what `Context.reconstruct` on a context of its own returns, bit for bit, in submission
order, whatever the depth, with frames of different size, dtype and parameters in flight at once, with failing frames in between.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_input, load_golden, load_points

pytestmark = pytest.mark.gpu


def series():
    """Eight frames: clouds of different size and shape (a dam break, a cube, the bunny), shifted from frame to frame, f32 and f64 mixed."""
    dam = load_points("double_dam_break_frame_26_4732_particles.npy").astype(np.float32)
    cube = golden_input(load_golden("cube_2366")).astype(np.float32)
    bunny = golden_input(load_golden("bunny_7705")).astype(np.float32)
    frames = []
    for k, base in enumerate([dam, cube, bunny, dam, cube, bunny, dam, cube]):
        pts = base + np.float32(0.003 * k) * np.array([1.0, -0.5, 0.25], np.float32)
        if k % 3 == 2:
            pts = pts[: len(pts) - 17 * k]
        frames.append(pts.astype(np.float64) if k in (3, 6) else pts)
    return frames


def mesh_of(res):
    v, t = res.mesh_views()
    return v.copy(), t.copy(), res.particle_densities.copy()


def same(a, b):
    return all(x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_frames_come_back_in_order_and_equal_the_one_context_loop(gpu_ctx, depth):
    from splashsurf_amd.api import FramePipeline, Parameters
    frames = series()
    prm = Parameters.new_relative(0.025, 4.0, 0.75, auto_disable=False, enable_simd=False)
    want, out = [], None
    for f in frames:  # the reference's loop: one workspace, one output object
        out = gpu_ctx.reconstruct(f, prm, out=out)
        want.append(mesh_of(out))
    with FramePipeline(0, depth) as pipe:
        got = [mesh_of(r) for r in pipe.map(frames, prm, fetch=FramePipeline.FETCH_VERTICES | FramePipeline.FETCH_TRIANGLES_U32 | FramePipeline.FETCH_DENSITIES)]
        assert pipe.in_flight == 0
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        assert same(a, b), "frame %d differs from the one-context loop" % k
    assert len({w[0].shape for w in want}) > 3  # (the frames really differ)


def test_tickets_capacity_and_result_lifetime(gpu_ctx):
    from splashsurf_amd.api import FramePipeline, Parameters, SplashsurfError
    frames = series()[:5]
    prm = Parameters.new_relative(0.025, 4.0, 1.0, auto_disable=False)
    with FramePipeline(0, 2) as pipe:
        assert pipe.depth == 2 and pipe.in_flight == 0 and not pipe.ready()
        with pytest.raises(SplashsurfError) as e:
            pipe.next()
        assert e.value.status == 6 and "no frame in flight" in str(e.value)
        assert pipe.submit(frames[0], prm) == 0
        assert pipe.submit(frames[1], prm) == 1
        with pytest.raises(SplashsurfError) as e:  # full: depth frames in flight
            pipe.submit(frames[2], prm)
        assert e.value.status == 6 and "full" in str(e.value)
        with pytest.raises(SplashsurfError):  # options only between frames
            pipe.set_two_pass(1)
        t0, r0 = pipe.next()
        assert t0 == 0 and pipe.in_flight == 1
        c0 = r0.counts()
        h0 = r0._h.value
        t1, r1 = pipe.next()
        assert t1 == 1 and r1._h.value != h0 and pipe.in_flight == 0
        assert r0.counts() == c0  # still valid: its slot was not resubmitted
        pipe.set_two_pass(1)
        assert pipe.submit(frames[2], prm) == 2
        t2, r2 = pipe.next()
        assert t2 == 2 and r2._h.value == h0  # slot 0 again: r0 IS r2 now
        assert r2.counts() == gpu_ctx.reconstruct(frames[2], prm).counts()
        rec_ms, fetch_ms = pipe.frame_times(0)
        assert rec_ms > 0.0 and fetch_ms >= 0.0


def test_a_failing_frame_reports_its_error_and_the_pipeline_goes_on(gpu_ctx):
    from splashsurf_amd.api import FramePipeline, GridConstructionError, Parameters, SplashsurfError
    frames = series()
    good = Parameters.new_relative(0.025, 4.0, 1.0, auto_disable=False)
    bad = Parameters.new_relative(0.025, 4.0, 1.0, auto_disable=False)
    bad.cube_size = 0.0  # the reference panics (density_map.rs:555-559): SS_ERR_UNKNOWN
    nan = frames[1].copy()
    nan[7, 1] = np.nan   # refused input (INTEGRATION.md difference 5)
    with FramePipeline(0, 3) as pipe:
        pipe.submit(frames[0], good)
        pipe.submit(frames[1], bad)
        pipe.submit(nan, good)
        _, r = pipe.next()
        first = mesh_of(r)
        with pytest.raises(SplashsurfError) as e:
            pipe.next()
        assert e.value.status == 4 and not isinstance(e.value, GridConstructionError)
        with pytest.raises(SplashsurfError) as e:
            pipe.next()
        assert e.value.status == 6 and "finite" in str(e.value)
        pipe.submit(frames[0], good)  # the slot of the failed frame is usable again
        _, r = pipe.next()
        assert same(mesh_of(r), first)
    want = gpu_ctx.reconstruct(frames[0], good)
    assert same(first, mesh_of(want))


def test_c_abi_argument_checks():
    from splashsurf_amd.api import load_library
    L = load_library()
    L.ss_pipeline_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    assert L.ss_pipeline_create(0, 0, C.byref(h)) == 6 and not h.value
    assert L.ss_pipeline_create(0, 9, C.byref(h)) == 6 and not h.value   # SS_PIPELINE_MAX_DEPTH = 8
    assert L.ss_pipeline_create(0, 1, None) == 6
    L.ss_pipeline_destroy.argtypes = [C.c_void_p]
    L.ss_pipeline_destroy.restype = None
    L.ss_pipeline_destroy(None)
    assert L.ss_pipeline_create(0, 2, C.byref(h)) == 0 and h.value
    L.ss_pipeline_submit_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
    assert L.ss_pipeline_submit_f32(h, None, 0, None, 0, None) == 6       # null parameters
    L.ss_pipeline_context.argtypes = [C.c_void_p, C.c_int]
    L.ss_pipeline_context.restype = C.c_void_p
    assert L.ss_pipeline_context(h, 0) and L.ss_pipeline_context(h, 1) and not L.ss_pipeline_context(h, 2)
    L.ss_pipeline_destroy(h)


def test_destroy_waits_for_the_frames_in_flight():
    from splashsurf_amd.api import FramePipeline, Parameters
    prm = Parameters.new_relative(0.025, 4.0, 0.75, auto_disable=False)
    pipe = FramePipeline(0, 2)
    for f in series()[:2]:
        pipe.submit(f, prm)
    pipe.close()  # frames never taken: destroy lets them finish, joins the threads, frees contexts and results
    assert pipe._h is None
