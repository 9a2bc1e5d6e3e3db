"""CPU twin of the splat's matrix-pipe certificate (round 6; runs without a GPU).

k_splat_fused certifies a 4^3 sub-block "inside the fluid" when a LOWER bound of the level set exceeds the threshold at all 64 points (ss_kernels.hip:
splat_cert_record / splat_cert_mfma / splat_cert_term4, make_device_params for the constants).  tests/test_gpu_certificates.py checks the property on the device; this
file restates the arithmetic of the certificate in numpy -- f16 operand records relative to the block's centre, the near lists by box distance, the K = 8 products
(exact in f32, summed here in f64 plus the slack the kernel reserves for the instruction's own accumulation), max(., 0)^4 -- and compares it with the ORACLE's exact
level set of the same subdomain (the reference's arithmetic, oracle/splash_oracle.c):

  * the bound never exceeds the exact value at any grid point (up to the 1e-4 relative margin the threshold carries), on fine and coarse grids and far from the origin;
  * hence every sub-block the restated certificate accepts holds only values above the threshold;
  * on bulk fluid it accepts a share of the sub-blocks comparable to the device's (the bound is useful, not merely valid).
"""
import numpy as np
import pytest

C4 = 0.76293          # SS_CERT_C4 (ss_kernels.hip), make_device_params: cert_vscale = C4 sigma (1 - 2e-5)
RNEAR = 0.64          # SS_TUNE_RNEAR (ss_api.hip)
f16, f32 = np.float16, np.float32


def _records(p, V, h, cs):
    """splat_cert_record for entries p (n, 3: relative to the block's centre, units of h, f32) with volumes V: the eight f16 slots per entry."""
    sig = f32(8.0) / (f32(np.pi) * (f32(h) * f32(h) * f32(h)))
    vscale = f32(C4 * float(sig) * (1.0 - 2.0e-5))
    xm = 3.5 * cs / h * (1.0 + 1.0e-5) + 1.0e-6
    r = 2.0 ** -10 * (1.0 + 2.0 ** -10)
    e1, e0 = f32(r * 2.0 * xm * (1.0 + 1.0e-6)), f32((r * 3.0 * xm * xm + 3.0e-5) * (1.0 + 1.0e-6))
    s = np.sqrt(np.sqrt((V.astype(f32) * vscale).astype(f32)).astype(f32)).astype(f32)
    eps = (e1 * ((np.abs(p[:, 0]) + np.abs(p[:, 1])) + np.abs(p[:, 2])) + e0).astype(f32)
    a = (((f32(1.0) - eps) - p[:, 0] * p[:, 0]) - (p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2])).astype(f32)
    sa = (s * a).astype(f32)
    sa_hi = sa.astype(f16)
    sa_lo = (sa - sa_hi.astype(f32)).astype(f32).astype(f16)
    P3 = ((s + s).astype(f32)[:, None] * p).astype(f32).astype(f16)
    ms = (-s).astype(f16)
    return sa_hi, sa_lo, P3, ms


def _tile_values(rec, x):
    """D[j, n] = sum over the eight slots of entry j and point n (x: points relative to the block's centre, units of h, f32)."""
    sa_hi, sa_lo, P3, ms = rec
    xh = x.astype(f16).astype(np.float64)
    xxh = (x * x).astype(f32).astype(f16).astype(np.float64)
    return (sa_hi.astype(np.float64)[:, None] + sa_lo.astype(np.float64)[:, None] + P3.astype(np.float64) @ xh.T + ms.astype(np.float64)[:, None] * xxh.sum(1)[None, :])


def _spline(q):
    return np.where(q < 0.5, 1.0 - 6.0 * q * q + 6.0 * q ** 3, np.where(q < 1.0, 2.0 * (1.0 - q) ** 3, 0.0))


CASES = [
    # name, tank scale, radius, cube size (radius-relative), offset in units of h
    ("fine_grid_bulk", 0.085, 0.005, 0.5, 0.0),
    ("far_from_origin", 0.07, 0.005, 0.5, 700.0),
    ("coarse_grid", 0.11, 0.005, 1.5, 0.0),
]


@pytest.mark.parametrize("name,scale,radius,cube,offset_h", CASES, ids=[c[0] for c in CASES])
def test_restated_certificate_bounds_the_oracles_level_set(oracle, name, scale, radius, cube, offset_h):
    from splashsurf_amd import workloads as W
    h = f32(4.0 * radius)
    cs = f32(cube * radius)
    pts = (W.tank_particles(scale, particle_radius=radius) + np.float32(offset_h * float(h))).astype(f32)
    par = oracle.make_params_relative(radius, 2.0, cube, subdomain_num_cubes_per_dim=64)
    orc = oracle.reconstruct_surface(pts, par)
    thr = 0.6
    mass = f32(1000.0) * (f32(2.0 * radius)) ** 3
    V = (mass / orc.particle_densities).astype(f32)
    gmin = orc.grid["aabb_min"].astype(f32)
    ns = [int(v) for v in orc.subdomain_grid["n_cells"]]
    sig = 8.0 / (np.pi * float(h) ** 3)
    # the occupied subdomain with the most fluid in it
    best, best_cnt = None, -1
    for flat in range(ns[0] * ns[1] * ns[2]):
        cnt, _ = oracle.levelset_subdomain(pts, par, flat)
        if cnt > best_cnt:
            best, best_cnt = flat, cnt
    cnt, G = oracle.levelset_subdomain(pts, par, best)
    s3 = (best // (ns[1] * ns[2]), (best // ns[2]) % ns[1], best % ns[2])
    rng = np.random.default_rng(3)
    n_sub = n_cert = 0
    worst_excess = -1.0
    blocks = [(bx, by, bz) for bx in range(8) for by in range(8) for bz in range(8)]
    rng.shuffle(blocks)
    inv_h = f32(1.0) / h
    for (bx, by, bz) in blocks[:40]:
        g0 = np.array([s3[0] * 64 + 8 * bx, s3[1] * 64 + 8 * by, s3[2] * 64 + 8 * bz])
        lo_blk = (gmin + g0.astype(f32) * cs).astype(f32)
        centre = (lo_blk + f32(3.5) * cs).astype(f32)
        # candidates: within the near radius of the block's box (k_splat_fused's near list); coordinates relative to the centre in units of h
        hi_blk = (gmin + (g0 + 7).astype(f32) * cs).astype(f32)
        e = np.maximum(np.maximum(lo_blk - pts, pts - hi_blk), 0.0)
        sel = np.nonzero((e * e).sum(1) <= (RNEAR * float(h)) ** 2)[0]
        if sel.size == 0:
            continue
        p_rel = ((pts[sel] - centre) * inv_h).astype(f32)
        rec = _records(p_rel, V[sel], float(h), float(cs))
        for sb in range(8):
            sx, sy, sz = (sb >> 2) & 1, (sb >> 1) & 1, sb & 1
            o = g0 + 4 * np.array([sx, sy, sz])
            ii, jj, kk = np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij")
            gp = o[None, :] + np.stack([ii.ravel(), jj.ravel(), kk.ravel()], 1)
            X = (gmin[None, :] + gp.astype(f32) * cs).astype(f32)                  # the points' coordinates as the scalar loop forms them
            slo, shi = X.min(0), X.max(0)
            es = np.maximum(np.maximum(slo - pts[sel], pts[sel] - shi), 0.0)
            near = np.nonzero((es * es).sum(1) <= (RNEAR * float(h)) ** 2)[0]
            x_rel = ((X - centre) * inv_h).astype(f32)
            D = _tile_values(tuple(r_[near] for r_ in rec), x_rel) + 1.0e-5       # (+ the room the kernel's slack leaves for the f32 accumulation inside the instruction)
            bound = (np.maximum(D, 0.0) ** 4).sum(0)
            # the exact level set at the same points: the oracle's own array of this subdomain (reference arithmetic) and, as a cross-check of the indexing, f64 sums
            loc = gp - np.array(s3) * 64
            exact = G[loc[:, 0], loc[:, 1], loc[:, 2]].astype(np.float64)
            if n_sub < 3:
                d = np.sqrt(((X.astype(np.float64)[:, None, :] - pts.astype(np.float64)[None, :, :]) ** 2).sum(2))
                ref64 = (V.astype(np.float64)[None, :] * sig * _spline(d / float(h))).sum(1)
                assert np.allclose(ref64, exact, rtol=2e-5, atol=1e-6)
            worst_excess = max(worst_excess, float(np.max(bound - exact * (1.0 + 1.0e-4))))
            assert np.all(bound <= exact * (1.0 + 1.0e-4) + 1e-12), (name, (bx, by, bz), sb, float(np.max(bound - exact)))
            n_sub += 1
            if np.all(bound > thr * 1.0001 * (1.0 + 192 * 1.2e-7)):
                n_cert += 1
                assert np.all(exact > thr)
    assert n_sub >= 50
    if name != "coarse_grid":  # bulk fluid on a fine grid: the restated certificate accepts a good share of the sub-blocks
        assert n_cert >= 0.25 * n_sub, (n_cert, n_sub)  # (a small tank is mostly surface: 39 % here, 88 % on S10M-tank on the device)
    assert worst_excess <= 0.0
