"""Randomised parity sweep (GPU vs oracle, bit for bit) over the parameter space the fixed cases do not reach:
cube radii R = ceil(h / cs) from 1 to 24, odd subdomain sizes, large coordinate offsets, negative coordinates, tiny and
huge length scales, anisotropic clouds, thresholds from 0.1 to 2, particle AABBs, both Real types and both
strategies.  Inputs are seeded; sizes keep the oracle in the sub-second range."""
import numpy as np
import pytest

from test_gpu_parity import assert_gpu_equals_oracle

pytestmark = pytest.mark.gpu


def _cloud(rng, n, kind, spacing):
    if kind == "lattice":  # exactly on a lattice: particles on cell / subdomain boundaries
        m = int(round(n ** (1.0 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[:n]
        return g.astype(np.float64) * spacing
    if kind == "jitter":
        m = int(round(n ** (1.0 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)[:n]
        return (g + rng.random(g.shape) * 0.6) * spacing
    if kind == "sheet":  # thin, anisotropic
        return rng.random((n, 3)) * np.array([40.0, 25.0, 2.0]) * spacing
    if kind == "clusters":
        c = rng.random((6, 3)) * 30.0 * spacing
        return c[rng.integers(0, 6, n)] + rng.normal(size=(n, 3)) * 1.5 * spacing
    raise KeyError(kind)


CASES = []
_rng = np.random.default_rng(20240926)
for i in range(36):
    scale = [1.0, 1e-3, 250.0][i % 3]                     # length scale of the whole problem
    r = 0.025 * scale
    c = [4.0, 2.0, 1.0, 0.75, 0.5, 0.33, 0.2, 1.3, 0.17][i % 9]   # cube size / radius  -> R = ceil(4 / c) in 1..24
    l = [2.0, 2.0, 1.5, 2.5][i % 4]
    n_cubes = [64, 16, 7, 32, 100, 9][i % 6]
    CASES.append(dict(
        seed=int(_rng.integers(1 << 30)), n=int([300, 1200, 2500, 60][i % 4] * (0.25 if c < 0.25 else 1.0)) + 3,
        kind=["jitter", "lattice", "sheet", "clusters"][(i // 2) % 4], r=r, l=l, c=c, n_cubes=n_cubes,
        t=[0.6, 0.1, 1.2, 0.6, 2.0][i % 5], offset=[0.0, -7.5, 1000.0, 0.0][i % 4] * scale,
        rest_density=[1000.0, 1.0, 650.0][i % 3], f64=(i % 5 == 4), strategy=["grid", "grid", "global"][i % 3], aabb=(i % 7 == 3)))


# the same sweep with the reference's default arithmetic (Parameters::enable_simd = true): GPU enable_simd = 1 against the
# oracle's uniform-SIMD mode 2, bit for bit (subdomain grid, f32 -- the only instantiation the reference's SIMD loop exists for)
SIMD_CASES = [dict(c, simd=1) for c in CASES if c["strategy"] == "grid" and not c["f64"]]
for c in CASES:
    c["simd"] = 0


@pytest.mark.parametrize("case", CASES + SIMD_CASES, ids=lambda c: "%s-%s-c%.2g-n%d-%s%s%s" % (c["kind"], c["strategy"], c["c"], c["n_cubes"], "f64" if c["f64"] else "f32",
                                                                                                "-aabb" if c["aabb"] else "", "-simd" if c["simd"] else ""))
def test_random_configuration_bit_identical(gpu_ctx, two_pass_ctx, oracle, case):
    _run_case(gpu_ctx, oracle, case)
    if case["strategy"] == "grid":  # the same with the two-pass splat forced on (certification of sub-blocks inside the fluid)
        res = _run_case(two_pass_ctx, oracle, case)
        assert res.stats["n_certified_subblocks"] >= 0


def _run_case(gpu_ctx, oracle, case):
    import splashsurf_amd as S
    dt = np.float64 if case["f64"] else np.float32
    rng = np.random.default_rng(case["seed"])
    pts = (_cloud(rng, case["n"], case["kind"], 2.0 * case["r"]) + case["offset"]).astype(np.float32).astype(dt)
    kw, okw = {}, {}
    if case["aabb"]:
        lo, hi = pts.min(axis=0), pts.max(axis=0)
        a, b = lo + 0.2 * (hi - lo), hi - 0.1 * (hi - lo)
        kw = dict(aabb_min=[float(x) for x in a], aabb_max=[float(x) for x in b])
        okw = dict(aabb_min=np.asarray(kw["aabb_min"], dt), aabb_max=np.asarray(kw["aabb_max"], dt))
    glob = case["strategy"] == "global"
    res = S.reconstruct_surface(pts, particle_radius=case["r"], rest_density=case["rest_density"], smoothing_length=case["l"], cube_size=case["c"],
                                iso_surface_threshold=case["t"], subdomain_grid=not glob, subdomain_grid_auto_disable=False,
                                subdomain_num_cubes_per_dim=case["n_cubes"], global_neighborhood_list=True, context=gpu_ctx, simd=case["simd"], **kw)
    par = oracle.make_params_relative(case["r"], case["l"], case["c"], iso_surface_threshold=case["t"], rest_density=case["rest_density"],
                                      subdomain_num_cubes_per_dim=case["n_cubes"], global_neighborhood_list=True, dtype=dt, subdomain_grid=not glob,
                                      simd=2 if case["simd"] else 0, **okw)
    orc = oracle.reconstruct_surface(pts, par)
    U = np.uint64 if case["f64"] else np.uint32
    if glob:
        assert res.subdomain_grid is None
        assert np.array_equal(res.particle_densities.view(U), orc.particle_densities.view(U))
        assert np.array_equal(res.vertex_keys, orc.vertex_keys)
        assert np.array_equal(res.mesh.vertices.view(U), orc.vertices.view(U))
        assert np.array_equal(res.mesh.triangles, orc.triangles)
    else:
        assert_gpu_equals_oracle(res, orc)
    ptr, idx = res.particle_neighbors_csr
    assert np.array_equal(ptr, orc.neighbor_ptr) and np.array_equal(idx, orc.neighbors)
    return res
