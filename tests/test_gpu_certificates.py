"""Soundness of the splat's certificates (-m gpu).

By default the level-set splat certifies 4^3 sub-blocks "inside the fluid" with a cheap LOWER bound of the level set (near particles only; since
round 6 the bound C4 u^4 below the cubic spline, evaluated for 32 entries x 32 points per v_mfma_f32_32x32x8_f16 on f16 operand records with a
slack that covers their rounding: ss_kernels.hip, splat_cert_record / splat_cert_mfma in k_splat_fused; over-dense blocks: the same tiles on
records relative to the sub-block's centre, k_splat_certify_big) and never evaluates them unless marching cubes reads their values.  The mesh tests show that the output does not change; this file checks the property
itself: for every sub-block that stayed certified, all 64 values of the COMPLETELY evaluated level set (SS_OPTION_FULL_LEVELSET on a second
context, bit-identical to the oracle: test_levelset_bit_identical_per_subdomain) lie above the iso-surface threshold -- on bulk fluid, on a scene
far from the origin (coordinate slack), on a coarse and on a fine grid (f16 ranges), and in both arithmetics.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name, tank scale, particle radius, cube size (radius-relative), offset (in units of h), simd
    ("tank_bulk_scalar", 0.22, 0.005, 0.5, 0.0, False),
    ("tank_bulk_simd", 0.22, 0.005, 0.5, 0.0, True),
    ("tank_far_from_origin", 0.16, 0.005, 0.5, 900.0, True),
    ("tank_coarse_grid", 0.30, 0.005, 1.5, 0.0, True),
    ("tank_fine_grid", 0.10, 0.005, 0.3, 0.0, False),
    ("tank_large_units", 0.16, 3.0, 0.5, -40.0, True),
]


@pytest.mark.parametrize("name,scale,radius,cube,offset_h,simd", CASES, ids=[c[0] for c in CASES])
def test_certified_subblocks_lie_inside_the_fluid(two_pass_ctx, full_levelset_ctx, name, scale, radius, cube, offset_h, simd):
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    h = 4.0 * radius
    pts = W.tank_particles(scale) * np.float32(radius / 0.005) + np.float32(offset_h * h)  # (the tank's geometry scaled with the radius)
    kw = dict(particle_radius=radius, smoothing_length=2.0, cube_size=cube, iso_surface_threshold=0.6, subdomain_grid=True, subdomain_grid_auto_disable=False, simd=simd)
    res = S.reconstruct_surface(pts, context=two_pass_ctx, **kw)
    masks, bxyz = res.certified_subblocks()
    n_cert = int(np.unpackbits(masks.view(np.uint8)).sum())
    assert masks.size == res.stats["n_active_blocks"]
    assert n_cert > 0.3 * 8 * masks.size, "the scene is meant to have bulk fluid: %d of %d sub-blocks certified" % (n_cert, 8 * masks.size)
    full = S.reconstruct_surface(pts, context=full_levelset_ctx, **kw)
    assert np.array_equal(full.mesh.vertices.view(np.uint32), res.mesh.vertices.view(np.uint32)) and np.array_equal(full.mesh.triangles_u32, res.mesh.triangles_u32)
    npnt = [int(x) for x in full.grid.npoints_per_dim]
    G = full.levelset_box([0, 0, 0], npnt)
    thr = np.float32(0.6)
    worst = np.inf
    checked = 0
    for sb in range(8):
        sel = np.nonzero((masks >> sb) & 1)[0]
        if not sel.size:
            continue
        o = bxyz[sel].astype(np.int64) * 8 + np.array([4 * ((sb >> 2) & 1), 4 * ((sb >> 1) & 1), 4 * (sb & 1)], dtype=np.int64)
        for dx in range(4):
            for dy in range(4):
                for dz in range(4):
                    x, y, z = o[:, 0] + dx, o[:, 1] + dy, o[:, 2] + dz
                    ok = (x < npnt[0]) & (y < npnt[1]) & (z < npnt[2])  # (points beyond the grid do not exist)
                    v = G[x[ok], y[ok], z[ok]]
                    worst = min(worst, float(v.min()))
                    checked += int(ok.sum())
    assert checked >= 60 * n_cert
    assert worst > float(thr), "a certified sub-block holds a level-set value of %.7g <= threshold %.7g" % (worst, float(thr))


def test_certificates_of_an_overwritten_call_are_refused(two_pass_ctx):
    """The masks live in the context's scratch: once another reconstruction has run on the context, asking an earlier result for them is an
    error, not garbage."""
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    pts = W.tank_particles(0.1)
    kw = dict(particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, iso_surface_threshold=0.6, subdomain_grid=True, subdomain_grid_auto_disable=False, simd=False)
    first = S.reconstruct_surface(pts, context=two_pass_ctx, **kw)
    masks, _ = first.certified_subblocks()
    assert masks.any()
    second = S.reconstruct_surface(pts[: pts.shape[0] // 2], context=two_pass_ctx, **kw)
    from splashsurf_amd.api import SplashsurfError
    with pytest.raises(SplashsurfError):
        first.certified_subblocks()
    masks2, _ = second.certified_subblocks()
    assert masks2.size == second.stats["n_active_blocks"]
