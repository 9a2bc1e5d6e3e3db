"""GPU parity tests of the reference's DEFAULT arithmetic, Parameters::enable_simd = true (-m gpu).

The reference's `simd=True` path mixes three arithmetics (AVX vector lanes, unfused remainder lanes, the scalar loop for
sparse subdomains; dense_subdomains.rs:991-1133, :1413-1415, :1590-1596), so its level-set values on a shared subdomain face
depend on the subdomain.  The HIP library evaluates every global grid point once with the AVX loop's arithmetic applied
uniformly (include/splashsurf_hip.h, enable_simd).  Three layers of evidence:

  1. enable_simd = 1 is BIT-IDENTICAL to the oracle's mode 2 (the same uniform arithmetic restated in C with fmaf and a
     correctly rounded sqrt): densities, level-set values, vertex coordinates, triangle sets;
  2. against the reference wheel's own simd=True output (tests/golden/simd_*.npz, tools/gen_goldens_simd.py): identical
     vertex-id / triangle sets and vertex coordinates within the north-star tolerance of 1e-5 relative (observed <= 1.3e-6;
     the reference's own scalar path is up to 9.6e-6 away from its SIMD path and differs in topology on config 5);
  3. enable_simd = 2 (hardware v_sqrt_f32, <= 1 ulp) stays within the reference's own assertion for its AVX loop --
     |difference to the scalar level set| < 100 f32::EPSILON (benches/bench_grid_loop.rs:254-260) -- and within 1e-5
     relative of mode 1 on vertex coordinates.
"""
import hashlib
import os

import numpy as np
import pytest

import mesh_compare as MC
from conftest import golden_input, golden_params, load_golden, device_name, device_sync
from test_gpu_parity import assert_gpu_equals_oracle

pytestmark = pytest.mark.gpu

SIMD_FULL = ["simd_kat1", "simd_cube_2366_n16", "simd_config1_double_dam_break"]
SIMD_DIGEST = ["simd_config1_n16", "simd_bunny_7705", "simd_config5_hilbert", "simd_tank_small", "simd_config2_s1m"]


def _run_gpu(ctx, pts, prm, simd):
    import splashsurf_amd as S
    return S.reconstruct_surface(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                 iso_surface_threshold=prm["iso_surface_threshold"], subdomain_grid=True, subdomain_grid_auto_disable=False,
                                 subdomain_num_cubes_per_dim=prm.get("subdomain_num_cubes_per_dim", 64), context=ctx, simd=simd)


def _oracle(O, pts, prm, simd):
    par = O.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                 subdomain_num_cubes_per_dim=prm.get("subdomain_num_cubes_per_dim", 64), simd=simd)
    return par, O.reconstruct_surface(pts, par)


@pytest.mark.parametrize("name", SIMD_FULL + SIMD_DIGEST)
def test_simd_bit_identical_to_uniform_oracle(gpu_ctx, oracle, name):
    g = load_golden(name)
    pts, prm = golden_input(g), golden_params(g)
    res = _run_gpu(gpu_ctx, pts, prm, True)
    assert res.stats["arith_mode"] in (2, 3)
    _, orc = _oracle(oracle, pts, prm, 2)
    assert_gpu_equals_oracle(res, orc)
    # densities do not depend on the flag, in the reference and here
    assert hashlib.sha256(res.particle_densities.tobytes()).hexdigest() == str(g["density_sha256"])


@pytest.mark.parametrize("name", SIMD_FULL)
def test_simd_matches_reference_simd_golden(gpu_ctx, name):
    g = load_golden(name)
    res = _run_gpu(gpu_ctx, golden_input(g), golden_params(g), True)
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"], cmp
    assert cmp["max_rel_diff"] <= 1e-5, cmp  # north_star tolerance; observed <= 1.3e-6
    assert MC.mesh_is_closed_manifold(res.mesh.triangles)


@pytest.mark.parametrize("name", SIMD_DIGEST)
def test_simd_matches_reference_simd_digest(gpu_ctx, name):
    g = load_golden(name)
    res = _run_gpu(gpu_ctx, golden_input(g), golden_params(g), True)
    ids, vs, tc = MC.canonicalize_geometric(res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert ids.size == int(g["n_vertices"]) and tc.shape[0] == int(g["n_triangles"])
    assert hashlib.sha256(ids.astype(np.int64).tobytes()).hexdigest() == str(g["ids_sha256"])
    assert hashlib.sha256(tc.astype(np.int64).tobytes()).hexdigest() == str(g["triangles_sha256"])
    sid, sv = g["sample_ids"], g["sample_vertices"].astype(np.float64)
    lo, hi = np.searchsorted(ids, sid, side="left"), np.searchsorted(ids, sid, side="right")
    assert np.all(hi > lo)
    single = (hi - lo) == 1
    worst = float(np.abs(vs[lo[single]].astype(np.float64) - sv[single]).max())
    for k in np.nonzero(~single)[0]:
        worst = max(worst, float(np.abs(vs[lo[k]:hi[k]].astype(np.float64) - sv[k]).max(axis=1).min()))
    assert worst <= 1e-5 * max(1.0, np.abs(sv).max()), worst
    assert MC.mesh_is_closed_manifold(res.mesh.triangles)


def _levelsets(res, oracle, pts, par):
    ns = res.subdomain_grid.ncells_per_dim
    n = 64
    for flat in range(ns[0] * ns[1] * ns[2]):
        cnt, ref = oracle.levelset_subdomain(pts, par, flat)
        if cnt < 0:
            continue
        s3 = (flat // (ns[1] * ns[2]), (flat // ns[2]) % ns[1], flat % ns[2])
        yield flat, ref, res.levelset_box([s3[0] * n, s3[1] * n, s3[2] * n], [n + 1] * 3)


def test_simd_levelset_bit_identical_per_subdomain(full_levelset_ctx, oracle):
    gpu_ctx = full_levelset_ctx
    """Every level-set value of config 1 (all four 65^3 subdomains, faces included) equals the uniform-SIMD oracle bit for bit."""
    g = load_golden("simd_config1_double_dam_break")
    pts, prm = golden_input(g), golden_params(g)
    res = _run_gpu(gpu_ctx, pts, prm, True)
    par, _ = _oracle(oracle, pts, prm, 2)
    seen = 0
    for flat, ref, got in _levelsets(res, oracle, pts, par):
        assert int((got.view(np.uint32) != ref.view(np.uint32)).sum()) == 0, flat
        seen += 1
    assert seen == 4


@pytest.mark.parametrize("name", ["simd_config1_double_dam_break", "simd_tank_small", "simd_config5_hilbert"])
def test_simd_hw_sqrt_within_the_references_own_tolerance(full_levelset_ctx, oracle, name):
    """enable_simd = 2 (v_sqrt_f32 instead of the correctly rounded root)."""
    gpu_ctx = full_levelset_ctx
    g = load_golden(name)
    pts, prm = golden_input(g), golden_params(g)
    res = _run_gpu(gpu_ctx, pts, prm, 2)
    assert res.stats["arith_mode"] == 4
    exact = _run_gpu(gpu_ctx, pts, prm, 1)
    # densities are untouched
    assert np.array_equal(res.particle_densities.view(np.uint32), exact.particle_densities.view(np.uint32))
    # level set: the reference asserts |avx - scalar| < 100 eps for its own SIMD loop (bench_grid_loop.rs:254-260)
    par0, _ = _oracle(oracle, pts, prm, 0)
    worst = 0.0
    for flat, ref, got in _levelsets(res, oracle, pts, par0):
        worst = max(worst, float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()))
    assert worst < 100 * np.finfo(np.float32).eps, worst
    # mesh: same topology as the exact-sqrt mode on these inputs, coordinates within the north-star tolerance
    cmp = MC.compare_keyed(res.mesh.vertices, res.vertex_keys, res.mesh.triangles, exact.mesh.vertices, exact.vertex_keys, exact.mesh.triangles)
    assert cmp["keys_equal"] and cmp["triangles_equal"], cmp
    order_a, order_b = np.argsort(res.vertex_keys, kind="stable"), np.argsort(exact.vertex_keys, kind="stable")
    va, vb = res.mesh.vertices[order_a].astype(np.float64), exact.mesh.vertices[order_b].astype(np.float64)
    assert np.abs(va - vb).max() <= 1e-5 * max(1.0, np.abs(vb).max())
    assert MC.mesh_is_closed_manifold(res.mesh.triangles)


def test_simd_on_reference_grid_loop_fixture(oracle):
    """The reference's captured input of its level-set loop (bench_grid_loop.rs): with the SIMD arithmetic the splat kernel
    satisfies the reference's own assertion |avx - scalar| < 100 eps, in both sqrt variants."""
    import torch
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, Parameters
    g = load_golden("grid_loop_subdomain_33_input")
    pts = np.ascontiguousarray(g["subdomain_particles"], dtype=np.float32)
    rho = np.ascontiguousarray(g["subdomain_particle_densities"], dtype=np.float32)
    h, cs, r = float(g["compact_support_radius"]), float(g["cell_size"]), 0.01
    gmin, nc = g["global_min"].astype(np.float64), g["global_n_points"] - 1
    margin = cs * np.ceil(np.float32(h) / np.float32(cs)) * (1 + np.sqrt(1.1920929e-07))
    dmin, dmax = gmin + r + margin + 0.5 * cs, gmin + nc * cs - r - margin - 0.5 * cs
    sub = [int(x) for x in g["subdomain_ijk"]]
    out = {}
    for simd in (0, 1, 2):
        ctx = Context(0)
        ctx.set_full_levelset(True)
        eng = D.HipEngine(ctx, Parameters(particle_radius=r, compact_support_radius=np.float32(h), cube_size=np.float32(cs), auto_disable=False, enable_simd=simd))
        shard = D.ShardDesc(dmin, dmax, sub, [s + 1 for s in sub])
        t = torch.from_numpy(pts).to(device_name())
        eng.begin(t, shard)
        res = eng.finish(torch.from_numpy(rho).to(device_name()))
        out[simd] = res.levelset_box([s * 64 for s in sub], [65] * 3).copy()
        eng.result._free()
        eng.ctx.close()
    assert out[0].max() > 0.6
    for simd in (1, 2):
        assert float(np.abs(out[simd].astype(np.float64) - out[0].astype(np.float64)).max()) < 100 * np.finfo(np.float32).eps


@pytest.mark.parametrize("simd", [0, 1, 2])
def test_early_exit_inside_the_fluid_changes_no_output(gpu_ctx, full_levelset_ctx, simd):
    """The default splat certifies 4^3 sub-blocks inside the fluid with a cheap lower bound (the sum over the nearby particles) and
    evaluates in full only what is not certified plus the certified blocks next to a sign change; SS_OPTION_FULL_LEVELSET evaluates
    everything.  Densities, vertices, triangles and
    edge keys must be identical bit for bit; the default run really did truncate blocks, and it refuses to hand out level-set
    values."""
    from splashsurf_amd import workloads as W
    pts = W.tank_particles(0.3)
    prm = dict(particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, iso_surface_threshold=0.6)
    a = _run_gpu(gpu_ctx, pts, prm, simd)
    b = _run_gpu(full_levelset_ctx, pts, prm, simd)
    sa, sb = a.stats, b.stats
    assert sb["n_truncated_blocks"] == 0 and sb["n_completed_blocks"] == 0
    assert sa["n_truncated_blocks"] + sa["n_completed_blocks"] > 0.3 * sa["n_active_blocks"], sa  # a bulk of fluid: many blocks are interior
    assert 0 < sa["n_completed_blocks"] < 0.5 * sa["n_active_blocks"], sa
    assert np.array_equal(a.particle_densities.view(np.uint32), b.particle_densities.view(np.uint32))
    assert np.array_equal(a.vertex_keys, b.vertex_keys)
    assert np.array_equal(a.mesh.vertices.view(np.uint32), b.mesh.vertices.view(np.uint32))
    assert np.array_equal(a.mesh.triangles_u32, b.mesh.triangles_u32)
    with pytest.raises(RuntimeError):
        a.levelset_box([0, 0, 0], [8, 8, 8])
    # ... and so does the C entry point itself (a C / C++ host does not go through the Python wrapper's check)
    import ctypes as C
    lo, ex, out = (C.c_int64 * 3)(0, 0, 0), (C.c_int64 * 3)(8, 8, 8), np.zeros(512, np.float32)
    assert a._lib.ss_result_levelset_box(a._h, lo, ex, out.ctypes.data_as(C.c_void_p)) == 6  # SS_ERR_INVALID_ARGUMENT
    assert b._lib.ss_result_levelset_box(b._h, lo, ex, out.ctypes.data_as(C.c_void_p)) == 0
    b.levelset_box([0, 0, 0], [8, 8, 8])
