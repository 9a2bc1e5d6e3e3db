"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle
(bit-exact) and against the reference's golden vectors (tests/golden/).

Tolerances: integer structure (vertex edge keys, triangle index sets) and all f32 values (densities,
level-set values, vertex coordinates) are required to be BIT-IDENTICAL to the oracle.  Against the
reference's own output the comparison allows 1e-5 relative on vertex coordinates (north_star), which
only absorbs the reference's order-dependent choice of coordinates for vertices on subdomain faces.
"""
import hashlib
import os

import numpy as np
import pytest

import mesh_compare as MC
from conftest import golden_input, golden_params, load_golden, device_name, device_sync

pytestmark = pytest.mark.gpu

FULL = ["kat1", "edge_empty", "edge_single", "edge_coincident", "edge_aabb_excludes_all", "cube_2366_aabb", "cube_8",
        "free_particles_125", "cube_2366", "bunny_7705", "config1_double_dam_break", "cube_2366_n16"]
DIGEST = ["config5_hilbert", "tank_small", "config2_s1m"]


def run_gpu(ctx, pts, prm, simd=False):
    import splashsurf_amd as S
    kw = {}
    if "aabb_min" in prm:
        kw = dict(aabb_min=prm["aabb_min"], aabb_max=prm["aabb_max"])
    return S.reconstruct_surface(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"],
                                 cube_size=prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                 subdomain_grid=True, subdomain_grid_auto_disable=False,
                                 subdomain_num_cubes_per_dim=prm.get("subdomain_num_cubes_per_dim", 64), context=ctx, simd=simd, **kw)


def run_oracle(O, pts, prm):
    kw = {}
    if "aabb_min" in prm:
        kw = dict(aabb_min=np.asarray(prm["aabb_min"], np.float32), aabb_max=np.asarray(prm["aabb_max"], np.float32))
    par = O.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                 iso_surface_threshold=prm["iso_surface_threshold"],
                                 subdomain_num_cubes_per_dim=prm.get("subdomain_num_cubes_per_dim", 64), **kw)
    return par, O.reconstruct_surface(pts, par)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def assert_gpu_equals_oracle(res, orc):
    g = res.grid
    assert list(g.ncells_per_dim) == list(orc.grid["n_cells"])
    assert g.aabb.min.dtype == orc.grid["aabb_min"].dtype
    assert np.array_equal(_bits(g.aabb.min), _bits(orc.grid["aabb_min"]))
    assert np.array_equal(_bits(g.aabb.max), _bits(orc.grid["aabb_max"]))
    sg = res.subdomain_grid
    assert list(sg.ncells_per_dim) == list(orc.subdomain_grid["n_cells"])
    rho = res.particle_densities
    assert rho.shape == orc.particle_densities.shape and rho.dtype == orc.particle_densities.dtype
    nbad = int((_bits(rho) != _bits(orc.particle_densities)).sum())
    assert nbad == 0, "%d of %d densities differ from the oracle" % (nbad, rho.size)
    cmp = MC.compare_keyed(res.mesh.vertices, res.vertex_keys, res.mesh.triangles, orc.vertices, orc.vertex_keys, orc.triangles)
    assert cmp["keys_equal"], cmp
    assert cmp["triangles_equal"], cmp
    assert cmp["vertices_bit_equal"], cmp
    assert res.subdomain_stats() == (orc.n_subdomains, orc.n_subdomain_particles)
    if orc.particle_inside_aabb is None:
        assert res.particle_inside_aabb is None
    else:
        assert np.array_equal(res.particle_inside_aabb, orc.particle_inside_aabb)


@pytest.mark.parametrize("name", FULL)
def test_gpu_bit_identical_to_oracle(gpu_ctx, oracle, name):
    g = load_golden(name)
    pts = golden_input(g)
    prm = golden_params(g)
    res = run_gpu(gpu_ctx, pts, prm)
    _, orc = run_oracle(oracle, pts, prm)
    assert_gpu_equals_oracle(res, orc)


@pytest.mark.parametrize("name", FULL)
def test_gpu_matches_reference_golden(gpu_ctx, name):
    g = load_golden(name)
    pts = golden_input(g)
    res = run_gpu(gpu_ctx, pts, golden_params(g))
    assert list(res.grid.ncells_per_dim) == list(g["n_cells"])
    assert np.array_equal(res.particle_densities.view(np.uint32), g["densities"].view(np.uint32))
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"], cmp
    assert cmp["max_rel_diff"] <= 1e-5, cmp  # north_star tolerance; observed: <= 1 ulp on subdomain-face vertices
    assert MC.mesh_is_closed_manifold(res.mesh.triangles)


@pytest.mark.parametrize("name", DIGEST)
def test_gpu_matches_reference_digest(gpu_ctx, name):
    g = load_golden(name)
    pts = golden_input(g)
    res = run_gpu(gpu_ctx, pts, golden_params(g))
    rho = res.particle_densities
    assert hashlib.sha256(rho.tobytes()).hexdigest() == str(g["density_sha256"]), "densities not bit-identical to the reference"
    ids, vs, tc = MC.canonicalize_geometric(res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert ids.size == int(g["n_vertices"]) and tc.shape[0] == int(g["n_triangles"])
    assert hashlib.sha256(ids.astype(np.int64).tobytes()).hexdigest() == str(g["ids_sha256"])
    assert hashlib.sha256(tc.astype(np.int64).tobytes()).hexdigest() == str(g["triangles_sha256"])
    # sampled reference vertices: nearest vertex of the same edge/grid-point cluster within 1e-5 relative
    sid, sv = g["sample_ids"], g["sample_vertices"].astype(np.float64)
    lo, hi = np.searchsorted(ids, sid, side="left"), np.searchsorted(ids, sid, side="right")
    assert np.all(hi > lo)
    worst = 0.0
    single = (hi - lo) == 1
    worst = max(worst, float(np.abs(vs[lo[single]].astype(np.float64) - sv[single]).max()))
    for k in np.nonzero(~single)[0]:
        worst = max(worst, float(np.abs(vs[lo[k]:hi[k]].astype(np.float64) - sv[k]).max(axis=1).min()))
    assert worst <= 1e-5 * max(1.0, np.abs(sv).max()), worst
    assert MC.mesh_is_closed_manifold(res.mesh.triangles)


@pytest.mark.parametrize("name", ["config5_hilbert", "tank_small"])
def test_gpu_bit_identical_to_oracle_large(gpu_ctx, oracle, name):
    g = load_golden(name)
    pts = golden_input(g)
    prm = golden_params(g)
    res = run_gpu(gpu_ctx, pts, prm)
    _, orc = run_oracle(oracle, pts, prm)
    assert_gpu_equals_oracle(res, orc)


@pytest.mark.parametrize("name", ["config1_double_dam_break", "cube_2366_n16", "tank_small"])
def test_levelset_bit_identical_per_subdomain(full_levelset_ctx, oracle, name):
    """The splat kernel alone: level-set values of every occupied subdomain (65^3 incl. shared faces)
    equal the oracle's density_grid_loop_scalar restatement bit for bit."""
    g = load_golden(name)
    pts = golden_input(g)
    prm = golden_params(g)
    par, _ = run_oracle(oracle, pts, prm)
    res = run_gpu(full_levelset_ctx, pts, prm)
    n = prm.get("subdomain_num_cubes_per_dim", 64)
    ns = res.subdomain_grid.ncells_per_dim
    checked = 0
    for flat in range(ns[0] * ns[1] * ns[2]):
        cnt, ref = oracle.levelset_subdomain(pts, par, flat)
        if cnt < 0:
            continue
        s = (flat // (ns[1] * ns[2]), (flat // ns[2]) % ns[1], flat % ns[2])
        got = res.levelset_box([s[0] * n, s[1] * n, s[2] * n], [n + 1] * 3)
        nbad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        assert nbad == 0, "subdomain %d: %d level-set values differ, max abs %g" % (flat, nbad, np.abs(got - ref).max())
        checked += 1
        if checked >= 12:
            break
    assert checked > 0


def test_device_pointer_input_and_determinism(gpu_ctx):
    """Input already resident in HBM (torch tensor) gives the same bits as host input; repeated calls are bit-stable."""
    import torch
    import splashsurf_amd as S
    pts = golden_input(load_golden("cube_2366"))
    a = S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, context=gpu_ctx)
    t = torch.from_numpy(pts).to(device_name())
    device_sync()
    b = S.reconstruct_surface(t, particle_radius=0.025, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, context=gpu_ctx)
    c = S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, context=gpu_ctx)
    for x in (b, c):
        assert np.array_equal(a.mesh.vertices.view(np.uint32), x.mesh.vertices.view(np.uint32))
        assert np.array_equal(a.mesh.triangles, x.mesh.triangles)
        assert np.array_equal(a.particle_densities.view(np.uint32), x.particle_densities.view(np.uint32))
    assert a.mesh.triangles.dtype == np.uint64 and a.mesh.vertices.dtype == np.float32
    assert np.array_equal(a.mesh.triangles.astype(np.uint32), a.mesh.triangles_u32)


def test_inplace_reuse(gpu_ctx, oracle):
    """reconstruct_surface_inplace semantics (lib.rs:340-346): the output object is cleared and reused."""
    import splashsurf_amd as S
    from splashsurf_amd.api import Parameters
    p1 = golden_input(load_golden("cube_2366"))
    p2 = golden_input(load_golden("cube_8"))
    prm = Parameters.new_relative(0.025, 4.0, 1.0, auto_disable=False)
    out = gpu_ctx.reconstruct(p1, prm)
    nv1 = out.counts()[0]
    gpu_ctx.reconstruct(p2, prm, out=out)
    fresh = gpu_ctx.reconstruct(p2, prm)
    assert out.counts() == fresh.counts() and out.counts()[0] != nv1
    assert np.array_equal(out.mesh.vertices.view(np.uint32), fresh.mesh.vertices.view(np.uint32))


def test_error_behaviour(gpu_ctx):
    import splashsurf_amd as S
    from splashsurf_amd.api import SplashsurfError
    pts = np.zeros((4, 3), np.float32)
    with pytest.raises(SplashsurfError) as e:
        S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.0, context=gpu_ctx)
    assert e.value.status == 4  # the reference panics (density_map.rs:555-559)
    with pytest.raises(TypeError):  # only float32 / float64 arrays, like pysplashsurf (reconstruction.rs:187-206)
        S.reconstruct_surface(pts.astype(np.float16), particle_radius=0.025, smoothing_length=2.0, cube_size=1.0, context=gpu_ctx)


def test_grid_for_reconstruction(gpu_ctx, oracle):
    import splashsurf_amd as S
    from splashsurf_amd.api import Parameters
    pts = golden_input(load_golden("bunny_7705"))
    g = S.grid_for_reconstruction(pts, Parameters.new_relative(0.025, 4.0, 0.75), context=gpu_ctx)
    o = oracle.grid_for_reconstruction(pts, oracle.make_params(0.025, np.float32(0.025) * np.float32(4.0), np.float32(0.025) * np.float32(0.75)))
    assert list(g.ncells_per_dim) == list(o["n_cells"])
    assert np.array_equal(g.aabb.min.view(np.uint32), o["aabb_min"].view(np.uint32))


@pytest.mark.parametrize("forced_two_pass", [False, True])
def test_dense_cloud_exceeding_tile_capacity(gpu_ctx, two_pass_ctx, oracle, forced_two_pass):
    """Over-dense input (many more candidates per level-set block than LDS slots): the multi-pass
    ordered accumulation must still be bit-identical to the oracle -- also with the certification scheme forced on (the
    lower-bound pass then runs over tiles of > 8192 entries that the gather kernel ordered in several passes)."""
    from splashsurf_amd import workloads as W
    if forced_two_pass:
        gpu_ctx = two_pass_ctx
    pts = (W.uniform_cube_particles(100000, seed=99) * np.float32(0.25)).astype(np.float32)
    prm = dict(particle_radius=0.01, smoothing_length=2.0, cube_size=1.0, iso_surface_threshold=0.6)
    # a block in the middle of the cloud sees the particles of a (7 cells + 2 x 4 cells)^3 box: more than two passes of
    # the 8192 index keys the large-tile kernel holds in LDS
    box = np.all(np.abs(pts - np.float32(0.125)) <= np.float32(0.075), axis=1).sum()
    assert box > 2 * 8192
    res = run_gpu(gpu_ctx, pts, prm)
    _, orc = run_oracle(oracle, pts, prm)
    assert res.stats["n_large_tile_blocks"] > 0 and res.stats["n_block_candidates"] > res.stats["n_active_blocks"] * 4096
    st = res.stats  # stage timers of the three splat kernels (HIP events on the library's stream)
    assert st["ms_levelset_gather"] > 0.0 and st["ms_levelset_accumulate"] > 0.0
    assert st["ms_levelset"] >= st["ms_levelset_gather"] + st["ms_levelset_accumulate"] - 1e-3  # (the stage is the sum of its parts up to rounding)
    assert_gpu_equals_oracle(res, orc)


@pytest.mark.parametrize("forced_two_pass", [False, True])
@pytest.mark.parametrize("simd", [False, True])
def test_dense_cloud_with_unordered_tiles(gpu_ctx, two_pass_ctx, oracle, forced_two_pass, simd):
    """Moderately over-dense input: tiles of a few hundred to a few thousand entries, which the large-tile gather leaves in scan
    order and the workgroup-level accumulate kernel orders in LDS when a block needs an exact sum."""
    from splashsurf_amd import workloads as W
    pts = (W.uniform_cube_particles(20000, seed=7) * np.float32(0.3)).astype(np.float32)
    prm = dict(particle_radius=0.01, smoothing_length=2.0, cube_size=1.0, iso_surface_threshold=0.6)
    res = run_gpu(two_pass_ctx if forced_two_pass else gpu_ctx, pts, prm, simd=simd)
    st = res.stats
    per_block = st["n_block_candidates"] / max(st["n_active_blocks"], 1)
    assert st["n_large_tile_blocks"] > 0 and 384 < per_block < 4096, (st["n_large_tile_blocks"], per_block)
    if forced_two_pass:
        assert st["n_certified_subblocks"] > 0
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                      simd=2 if simd else 0)
    orc = oracle.reconstruct_surface(pts, par)
    assert_gpu_equals_oracle(res, orc)


@pytest.mark.parametrize("case", [("double_dam_break_frame_26_4732_particles.npy", 0.025, 2.0, 1.1, 16, 2, np.float32),
                                  ("hilbert_46843_particles.npy", 0.025, 2.0, 0.9, 32, 3, np.float32),
                                  ("double_dam_break_frame_26_4732_particles.npy", 0.025, 2.0, 1.1, 16, 3, np.float64)])
def test_sharded_engine_reproduces_full_reconstruction(oracle, case):
    """The multi-GPU path on ONE GPU: k pseudo-ranks (one HIP context each) reconstruct slabs of subdomains
    through ss_shard_begin_f32 / ss_shard_finish with the density exchange done by hand; the merged
    result must equal the single-process oracle bit for bit."""
    import torch
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, Parameters
    fn, r, l, c, n_cubes, k, dt = case
    U = np.uint32 if dt == np.float32 else np.uint64
    pts = np.load(os.path.join(os.path.dirname(__file__), "data", fn)).astype(dt)
    prm = Parameters(particle_radius=r, compact_support_radius=dt(2.0 * l * r), cube_size=dt(c * r),
                     subdomain_num_cubes_per_dim=n_cubes, auto_disable=False, enable_simd=False)
    engines = [D.HipEngine(Context(0), prm, dtype=dt) for _ in range(k)]
    P_all = torch.from_numpy(pts).to(device_name())
    dmin, dmax = pts.min(axis=0), pts.max(axis=0)
    gmin, sub_size, ns, margin, _ = engines[0].grid_for_domain(dmin, dmax)
    axis = int(np.argmax(ns))
    slabs = D.partition_slabs(P_all[:, axis], float(gmin[axis]), sub_size, ns[axis], k)
    assert sum(1 for lo, hi in slabs if hi > lo) >= 2
    rho_global = torch.zeros(pts.shape[0], dtype=torch.float32 if dt == np.float32 else torch.float64, device=device_name())
    sel = []
    for q, (lo, hi) in enumerate(slabs):
        sub_lo, sub_hi = [0, 0, 0], list(ns)
        sub_lo[axis], sub_hi[axis] = lo, hi
        shard = D.ShardDesc(dmin, dmax, sub_lo, sub_hi)
        pad = margin * 1.001 + 1e-6
        c_lo, c_hi = float(gmin[axis]) + lo * sub_size - pad, float(gmin[axis]) + hi * sub_size + pad
        ids = torch.nonzero((P_all[:, axis] >= c_lo) & (P_all[:, axis] <= c_hi), as_tuple=False).squeeze(1) if hi > lo else torch.zeros(0, dtype=torch.int64, device=device_name())
        L = P_all.index_select(0, ids).contiguous()
        rho_local = engines[q].begin(L, shard)
        rho_global.index_add_(0, ids, rho_local)  # stands in for the all-reduce: one non-zero contribution per particle
        sel.append(ids)
    V, K, T = [], [], []
    voff = 0
    for q in range(k):
        res = engines[q].finish(rho_global.index_select(0, sel[q]).contiguous())
        V.append(res.mesh.vertices)
        K.append(res.vertex_keys)
        T.append(res.mesh.triangles.astype(np.int64) + voff)
        voff += res.mesh.vertices.shape[0]
    V, K, T = np.concatenate(V), np.concatenate(K), np.concatenate(T)
    uk, first = np.unique(K, return_index=True)
    merged_v, merged_t = V[first], np.searchsorted(uk, K)[T]
    # duplicates on slab faces are bit-identical
    order = np.argsort(K, kind="stable")
    same = K[order][1:] == K[order][:-1]
    assert same.sum() > 0
    assert np.array_equal(V[order][1:][same].view(U), V[order][:-1][same].view(U))
    ref = oracle.reconstruct_surface(pts, oracle.make_params(r, dt(2.0 * l * r), dt(c * r), subdomain_num_cubes_per_dim=n_cubes, dtype=dt))
    assert np.array_equal(rho_global.cpu().numpy().view(U), ref.particle_densities.view(U))
    cmp = MC.compare_keyed(merged_v, uk, merged_t, ref.vertices, ref.vertex_keys, ref.triangles)
    assert cmp["keys_equal"] and cmp["triangles_equal"] and cmp["vertices_bit_equal"], cmp


def _pseudo_rank_bricks(pts, prm, k, dt=np.float32):
    """k pseudo-ranks (one HIP context each, all on GPU 0) reconstruct the bricks `bricks_from_histogram` cuts for k ranks,
    through ss_shard_begin / ss_shard_finish with the density exchange done by hand.  Returns the merged mesh
    (vertices, keys, triangles), the global density vector and the bricks."""
    import torch
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context
    engines = [D.HipEngine(Context(0), prm, dtype=dt) for _ in range(k)]
    P_all = torch.from_numpy(pts).to(device_name())
    dmin, dmax = pts.min(axis=0), pts.max(axis=0)
    gmin, sub_size, ns, margin, _ = engines[0].grid_for_domain(dmin, dmax)
    sub = [torch.floor((P_all[:, d] - float(gmin[d])) / sub_size).to(torch.int64).clamp_(0, ns[d] - 1) for d in range(3)]
    hist3 = torch.bincount((sub[0] * ns[1] + sub[1]) * ns[2] + sub[2], minlength=ns[0] * ns[1] * ns[2]).cpu().numpy().reshape(ns)
    bricks = D.bricks_from_histogram(hist3, k)
    rho_global = torch.zeros(pts.shape[0], dtype=torch.float32 if dt == np.float32 else torch.float64, device=device_name())
    sel = []
    pad = margin * 1.001 + 1e-6
    for q, (lo, hi) in enumerate(bricks):
        m = torch.ones(P_all.shape[0], dtype=torch.bool, device=device_name())
        for d in range(3):
            m &= (P_all[:, d] >= float(gmin[d]) + lo[d] * sub_size - pad) & (P_all[:, d] <= float(gmin[d]) + hi[d] * sub_size + pad)
        if any(hi[d] <= lo[d] for d in range(3)):
            m &= False
        ids = torch.nonzero(m, as_tuple=False).squeeze(1)
        L = P_all.index_select(0, ids).contiguous()
        rho_local = engines[q].begin(L, D.ShardDesc(dmin, dmax, lo, hi))
        rho_global.index_add_(0, ids, rho_local)  # stands in for the exchange: one non-zero contribution per particle
        sel.append(ids)
    V, K, T = [], [], []
    voff = 0
    for q in range(k):
        res = engines[q].finish(rho_global.index_select(0, sel[q]).contiguous())
        V.append(res.mesh.vertices)
        K.append(res.vertex_keys)
        T.append(res.mesh.triangles.astype(np.int64) + voff)
        voff += res.mesh.vertices.shape[0]
    for e in engines:
        e.result._free()
        e.ctx.close()
    return np.concatenate(V), np.concatenate(K), np.concatenate(T), rho_global.cpu().numpy(), bricks


def test_config4_s40m_tank(gpu_ctx):
    """BASELINE config 4 (S40M-tank, 39.8 M particles): (a) the full cloud once on ONE GPU -- closed manifold mesh, unique
    edge keys, every density positive; (b) a 1.24 M-particle crop of the same tank cut into the 4 and 8 bricks the
    multi-GPU path uses (pseudo-ranks on one device through the shard ABI) equals the direct reconstruction bit for bit."""
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Parameters
    wl = W.WORKLOADS["s40m_tank"]
    kw = dict(particle_radius=wl["particle_radius"], smoothing_length=wl["smoothing_length"], cube_size=wl["cube_size"], iso_surface_threshold=0.6)
    pts = wl["gen"]()
    assert pts.shape[0] > 39_000_000
    res = run_gpu(gpu_ctx, pts, kw)
    nv, nt = res.counts()
    assert nv > 15_000_000 and nt > 30_000_000
    assert res.stats["n_large_tile_blocks"] == 0
    tri = res.mesh.triangles_u32
    assert MC.mesh_is_closed_manifold(tri)
    keys = res.vertex_keys
    assert np.unique(keys).size == keys.size
    rho = res.particle_densities
    assert rho.shape[0] == pts.shape[0] and float(rho.min()) > 0.0
    del res, tri, keys, rho, pts
    # (b) bricks == direct on a >= 1 M crop
    crop = W.tank_particles(0.5)
    assert crop.shape[0] >= 1_000_000
    direct = run_gpu(gpu_ctx, crop, kw)
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                     auto_disable=False, enable_simd=False)
    for k in (4, 8):
        V, K, T, rho_g, bricks = _pseudo_rank_bricks(crop, prm, k)
        assert sum(1 for lo, hi in bricks if all(hi[d] > lo[d] for d in range(3))) == k
        assert np.array_equal(rho_g.view(np.uint32), direct.particle_densities.view(np.uint32))
        uk, first = np.unique(K, return_index=True)
        cmp = MC.compare_keyed(V[first], uk, np.searchsorted(uk, K)[T], direct.mesh.vertices, direct.vertex_keys, direct.mesh.triangles)
        assert cmp["keys_equal"] and cmp["triangles_equal"] and cmp["vertices_bit_equal"], cmp


@pytest.mark.parametrize("simd", [False, True], ids=["scalar", "simd"])
def test_full_size_s10m_tank_bit_identical_to_oracle(gpu_ctx, oracle, simd):
    """BASELINE config 3 at FULL size (10 M particles, ~2.3 G grid cells): the whole result -- 10 M
    densities, 7.2 M vertices, 14.4 M triangles -- equals the CPU oracle bit for bit; plus the
    size-independent properties the reference's tests assert (closed manifold mesh) and run-to-run
    determinism.  Both arithmetics: enable_simd = 0 against the oracle's scalar loop (dense_subdomains.rs:784-847)
    and enable_simd = 1 -- the mode bench.py measures -- against the oracle's uniform AVX arithmetic (mode 2,
    dense_subdomains.rs:991-1133)."""
    from splashsurf_amd import workloads as W
    wl = W.WORKLOADS["s10m_tank"]
    pts = wl["gen"]()
    prm = dict(particle_radius=wl["particle_radius"], smoothing_length=wl["smoothing_length"], cube_size=wl["cube_size"], iso_surface_threshold=0.6)
    res = run_gpu(gpu_ctx, pts, prm, simd=simd)
    assert res.stats["arith_mode"] in ((2, 3) if simd else (0, 1))
    nv, nt = res.counts()
    assert nv > 5_000_000 and nt > 10_000_000
    h1 = hashlib.sha256(res.mesh.vertices.tobytes() + res.mesh.triangles_u32.tobytes() + res.particle_densities.tobytes()).hexdigest()
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], iso_surface_threshold=0.6,
                                      subdomain_num_cubes_per_dim=64, simd=2 if simd else 0)
    orc = oracle.reconstruct_surface(pts, par)
    assert_gpu_equals_oracle(res, orc)
    assert MC.mesh_is_closed_manifold(res.mesh.triangles_u32)
    keys = res.vertex_keys
    assert np.unique(keys).size == keys.size
    res2 = run_gpu(gpu_ctx, pts, prm, simd=simd)
    h2 = hashlib.sha256(res2.mesh.vertices.tobytes() + res2.mesh.triangles_u32.tobytes() + res2.particle_densities.tobytes()).hexdigest()
    assert h1 == h2


@pytest.mark.parametrize("name", ["neighbors_cube_2366_n16", "neighbors_config1"])
def test_gpu_neighbor_lists_match_reference(gpu_ctx, name):
    import splashsurf_amd as S
    g = load_golden(name)
    prm = golden_params(g)
    res = S.reconstruct_surface(golden_input(g), particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"],
                                cube_size=prm["cube_size"], subdomain_grid_auto_disable=False,
                                subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], global_neighborhood_list=True, context=gpu_ctx)
    row, idx = res.particle_neighbors_csr
    assert np.array_equal(row.astype(np.int64), g["row_ptr"])
    assert np.array_equal(idx.astype(np.int64), g["neighbors"].astype(np.int64))
    lists = res.particle_neighbors
    assert len(lists) == g["row_ptr"].size - 1 and lists[0].dtype == np.uint64
    # without the flag the attribute is None (Option::None, lib.rs:57-60)
    res2 = S.reconstruct_surface(golden_input(g), particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"],
                                 cube_size=prm["cube_size"], subdomain_grid_auto_disable=False, context=gpu_ctx)
    assert res2.particle_neighbors is None


def test_splat_on_reference_grid_loop_fixture(oracle):
    """The reference's own captured input of its level-set loop (data/density_grid_loop_subdomain_33.json,
    used by benches/bench_grid_loop.rs:254-260 to assert AVX == scalar within 100 eps): 6 821 particles
    WITH their densities and a 65^3 grid.  The splat kernel alone (densities injected through the shard
    API) must equal density_grid_loop_scalar -- here bit for bit, not within 100 eps."""
    import torch
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, Parameters
    g = load_golden("grid_loop_subdomain_33_input")
    pts = np.ascontiguousarray(g["subdomain_particles"], dtype=np.float32)
    rho = np.ascontiguousarray(g["subdomain_particle_densities"], dtype=np.float32)
    h, cs, r = float(g["compact_support_radius"]), float(g["cell_size"]), 0.01
    d = np.float32(r) + np.float32(r)
    assert (d * d * d) * np.float32(1000.0) == g["particle_rest_mass"]  # kernel.rs:28-30, dense_subdomains.rs:117-118
    gmin, nc = g["global_min"].astype(np.float64), g["global_n_points"] - 1
    margin = cs * np.ceil(np.float32(h) / np.float32(cs)) * (1 + np.sqrt(1.1920929e-07))
    dmin, dmax = gmin + r + margin + 0.5 * cs, gmin + nc * cs - r - margin - 0.5 * cs
    opar = oracle.make_params(r, h, cs)
    G, SG, _ = oracle.grid_for_domain(opar, dmin, dmax)
    assert np.array_equal(G["aabb_min"].view(np.uint32), g["global_min"].view(np.uint32)) and list(G["n_points"]) == list(g["global_n_points"])
    sub = [int(x) for x in g["subdomain_ijk"]]
    ns = [int(x) for x in SG["n_cells"]]
    flat = (sub[0] * ns[1] + sub[1]) * ns[2] + sub[2]
    cnt, ref = oracle.shard_levelset(pts, rho, opar, dmin, dmax, sub, [s + 1 for s in sub], flat)
    assert cnt == pts.shape[0]  # every particle of the fixture is a member of that subdomain
    ctx = Context(0)
    ctx.set_full_levelset(True)
    eng = D.HipEngine(ctx, Parameters(particle_radius=r, compact_support_radius=np.float32(h), cube_size=np.float32(cs), auto_disable=False, enable_simd=False))
    shard = D.ShardDesc(dmin, dmax, sub, [s + 1 for s in sub])
    eng.begin(torch.from_numpy(pts).to(device_name()), shard)
    res = eng.finish(torch.from_numpy(rho).to(device_name()))
    got = res.levelset_box([s * 64 for s in sub], [65, 65, 65])
    assert int((ref != 0).sum()) > 100000
    nbad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    assert nbad == 0, "%d of %d level-set values differ (max abs %g)" % (nbad, ref.size, np.abs(got - ref).max())


def test_cpp_host_over_c_abi(tmp_path):
    """A C++ host (include/splashsurf_hip.hpp, mirroring the reference's Rust API) drives the C ABI without
    Python: the reference's known-answer test (test_simple.rs:71-126), in-place reuse, neighbour lists, the
    error variants, the one-rank sharded path and a time series through FrameSeries (ss_pipeline_*)."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "test_host")
    # (the library under test: the HIP build, or the CPU execution model of tests/emu when SPLASHSURF_HIP_LIB names it -- tests/test_emu_kernels.py)
    lib = os.environ.get("SPLASHSURF_HIP_LIB") or os.path.join(root, "splashsurf_amd", "libsplashsurf_hip.so")
    libdir, libname = os.path.dirname(os.path.abspath(lib)), os.path.basename(lib)[3:-3]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "test_host.cpp"),
                           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


F64 = ["f64_kat1", "f64_cube_2366_n16", "f64_free_particles_125", "f64_config1", "f64_tank_small"]


@pytest.mark.parametrize("name", F64)
def test_gpu_f64_bit_identical_to_oracle_and_reference(gpu_ctx, oracle, name):
    """reconstruct_surface::<i64, f64>: float64 arrays take the f64 instantiation of every kernel; densities,
    vertices (64-bit patterns) and triangle sets equal the f64 oracle, which is pinned to the reference's f64 path."""
    import splashsurf_amd as S
    g = load_golden(name)
    prm = golden_params(g)
    pts = golden_input(g).astype(np.float64)
    res = S.reconstruct_surface(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                iso_surface_threshold=prm["iso_surface_threshold"], subdomain_grid_auto_disable=False,
                                subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], context=gpu_ctx)
    assert res.is_f64 and res.mesh.vertices.dtype == np.float64 and res.particle_densities.dtype == np.float64
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                      iso_surface_threshold=prm["iso_surface_threshold"],
                                      subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], dtype=np.float64)
    orc = oracle.reconstruct_surface(pts, par)
    assert_gpu_equals_oracle(res, orc)
    # and the reference's own f64 output
    assert np.array_equal(res.particle_densities.view(np.uint64), g["densities"].view(np.uint64))
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] <= 1e-13, cmp
    # f32 results are still f32 afterwards (buffers are shared between the instantiations)
    res32 = S.reconstruct_surface(pts.astype(np.float32), particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"],
                                  cube_size=prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                  subdomain_grid_auto_disable=False, subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], context=gpu_ctx)
    assert not res32.is_f64 and res32.mesh.vertices.dtype == np.float32


def test_gpu_f64_levelset_bit_identical(full_levelset_ctx, oracle):
    gpu_ctx = full_levelset_ctx
    import splashsurf_amd as S
    g = load_golden("f64_config1")
    prm = golden_params(g)
    pts = golden_input(g).astype(np.float64)
    res = S.reconstruct_surface(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                subdomain_grid_auto_disable=False, context=gpu_ctx)
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], dtype=np.float64)
    ns = res.subdomain_grid.ncells_per_dim
    for flat in range(ns[0] * ns[1] * ns[2]):
        cnt, ref = oracle.levelset_subdomain(pts, par, flat)
        if cnt < 0:
            continue
        s3 = (flat // (ns[1] * ns[2]), (flat // ns[2]) % ns[1], flat % ns[2])
        got = res.levelset_box([s3[0] * 64, s3[1] * 64, s3[2] * 64], [65] * 3)
        assert got.dtype == np.float64
        assert int((got.view(np.uint64) != ref.view(np.uint64)).sum()) == 0


# ---- global (non-decomposed) strategy, SURVEY rows A14/A15 ----
GLOBAL = ["global_kat1", "global_edge_empty", "global_cube_8", "global_cube_2366", "global_cube_2366_auto_disable",
          "global_cube_2366_aabb", "global_free_particles_125", "global_config1", "global_f64_cube_2366", "global_f64_config1"]


def _run_global(gpu_ctx, oracle, g):
    import splashsurf_amd as S
    prm = golden_params(g)
    dt = g["densities"].dtype.type
    pts = golden_input(g).astype(dt)
    kw = {}
    if "aabb_min" in prm:
        kw = dict(aabb_min=prm["aabb_min"], aabb_max=prm["aabb_max"])
    res = S.reconstruct_surface(pts, particle_radius=prm["particle_radius"], smoothing_length=prm["smoothing_length"], cube_size=prm["cube_size"],
                                iso_surface_threshold=prm["iso_surface_threshold"], subdomain_grid=prm.get("subdomain_grid", True),
                                subdomain_grid_auto_disable=prm.get("subdomain_grid_auto_disable", False), context=gpu_ctx, **kw)
    okw = {k: np.asarray(v, dt) for k, v in kw.items()}
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                      dtype=dt, subdomain_grid=prm.get("subdomain_grid", True),
                                      subdomain_grid_auto_disable=prm.get("subdomain_grid_auto_disable", False), **okw)
    return res, oracle.reconstruct_surface(pts, par), dt


@pytest.mark.gpu
@pytest.mark.parametrize("name", GLOBAL)
def test_gpu_global_strategy_bit_identical_to_oracle_and_reference(gpu_ctx, oracle, name):
    """reconstruct_surface_global on the GPU: densities, neighbour lists, level-set values, vertices (coordinates AND
    order = ascending edge key) and triangles are bit-identical to the oracle, hence to the reference's sequential path
    (the goldens come from the reference itself and are checked here as well)."""
    g = load_golden(name)
    res, orc, dt = _run_global(gpu_ctx, oracle, g)
    U = np.uint32 if dt == np.float32 else np.uint64
    assert orc.used_global_strategy and res.subdomain_grid is None
    assert list(res.grid.ncells_per_dim) == list(g["n_cells"])
    assert np.array_equal(np.asarray(res.grid.aabb.min, dtype=dt).view(U), g["grid_min"].view(U))
    assert np.array_equal(np.asarray(res.grid.aabb.max, dtype=dt).view(U), g["grid_max"].view(U))
    # reference goldens
    assert np.array_equal(res.particle_densities.view(U), g["densities"].view(U))
    ptr, idx = res.particle_neighbors_csr
    assert np.array_equal(ptr.astype(np.int64), g["row_ptr"]) and np.array_equal(idx.astype(np.int64), g["neighbors"].astype(np.int64))
    if "inside" in g.files:
        assert np.array_equal(res.particle_inside_aabb, g["inside"].astype(bool))
    # oracle: everything bit-identical including order
    assert np.array_equal(res.vertex_keys, orc.vertex_keys)
    assert np.array_equal(res.mesh.vertices.view(U), orc.vertices.view(U))
    assert np.array_equal(res.mesh.triangles, orc.triangles)
    if orc.global_levelset is not None:
        npts = [int(x) for x in g["n_points"]]
        got = res.levelset_box([0, 0, 0], npts)
        assert int((got.view(U) != orc.global_levelset.view(U)).sum()) == 0
    # the reference's mesh (hash-map order) under geometric canonicalisation
    if g["vertices"].shape[0]:
        cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.mesh.vertices, res.mesh.triangles, g["grid_min"], g["cell_size"], g["n_points"])
        assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0, cmp
    else:
        assert res.mesh.vertices.shape[0] == 0 and res.mesh.triangles.shape[0] == 0


@pytest.mark.gpu
def test_gpu_auto_disable_rule(gpu_ctx):
    """lib.rs:421-441: decomposition iff max cells per dim > (1.2 n) as u32; default parameters follow the reference (auto_disable on)."""
    import splashsurf_amd as S
    pts = golden_input(load_golden("global_cube_2366"))
    for n_cubes, expect_global in ((64, True), (48, True), (41, False), (16, False)):
        res = S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.75, subdomain_num_cubes_per_dim=n_cubes, context=gpu_ctx)
        assert (res.subdomain_grid is None) == expect_global, n_cubes
        assert MC.mesh_is_closed_manifold(res.mesh.triangles)


@pytest.mark.gpu
def test_gpu_global_strategy_medium_input(gpu_ctx, oracle):
    """SpatialDecomposition::None on a domain well beyond the auto-disable size (bunny, 127x130x154 cells)."""
    import splashsurf_amd as S
    pts = golden_input(load_golden("bunny_7705"))
    res = S.reconstruct_surface(pts, particle_radius=0.025, smoothing_length=2.0, cube_size=0.5, subdomain_grid=False, context=gpu_ctx)
    orc = oracle.reconstruct_surface(pts, oracle.make_params_relative(0.025, 2.0, 0.5, subdomain_grid=False))
    assert np.array_equal(res.particle_densities.view(np.uint32), orc.particle_densities.view(np.uint32))
    assert np.array_equal(res.vertex_keys, orc.vertex_keys)
    assert np.array_equal(res.mesh.vertices.view(np.uint32), orc.vertices.view(np.uint32))
    assert np.array_equal(res.mesh.triangles, orc.triangles)
    assert res.mesh.vertices.shape[0] == 73638


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["host", "hbm"])
def test_gpu_dense_marching_cubes(gpu_ctx, oracle, where):
    """ss_marching_cubes_* (pysplashsurf.marching_cubes): identical to the oracle incl. order, to the reference's mesh
    under canonicalisation; the reference's triangulation error is reported as SS_ERR_MARCHING_CUBES."""
    import torch
    import splashsurf_amd as S
    from splashsurf_amd.api import SplashsurfError
    g = load_golden("marching_cubes_dense")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        U = np.uint32 if dt == np.float32 else np.uint64
        vals = np.ascontiguousarray(g["values"].astype(dt))
        arg = torch.from_numpy(vals).to(device_name()) if where == "hbm" else vals
        mesh, grid = S.marching_cubes(arg, iso_surface_threshold=float(g["threshold"]), cube_size=float(g["cube_size"]), translation=list(g["translation"]),
                                      return_grid=True, context=gpu_ctx)
        orc = oracle.marching_cubes(vals, float(g["threshold"]), float(g["cube_size"]), g["translation"])
        assert list(grid.npoints_per_dim) == list(vals.shape) and np.array_equal(np.asarray(grid.aabb.min, dtype=dt), g["translation"].astype(dt))
        assert mesh.vertices.dtype == dt
        assert np.array_equal(mesh.vertices.view(U), orc.vertices.view(U)) and np.array_equal(mesh.triangles, orc.triangles)
        cmp = MC.compare_geometric(g["v_" + tag], g["t_" + tag].astype(np.int64), mesh.vertices, mesh.triangles, orc.grid["aabb_min"], dt(g["cube_size"]),
                                   np.array(vals.shape))
        assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0, cmp
    with pytest.raises(SplashsurfError) as e:
        S.marching_cubes(g["eq_values"], iso_surface_threshold=1.0, cube_size=1.0, context=gpu_ctx)
    assert e.value.status == 3
    with pytest.raises(SplashsurfError):
        S.marching_cubes(np.zeros((1, 4, 4), np.float32), iso_surface_threshold=0.5, cube_size=1.0, context=gpu_ctx)


@pytest.mark.gpu
def test_gpu_neighborhood_search_standalone(gpu_ctx):
    """ss_neighborhood_search_*: the lists of the reference's sequential search (golden from the reference's global path)."""
    import splashsurf_amd as S
    for name in ("global_cube_2366", "global_f64_cube_2366"):
        g = load_golden(name)
        dt = g["densities"].dtype.type
        pts = golden_input(g).astype(dt)
        h = dt(2.0 * 2.0 * 0.025)
        nl = S.neighborhood_search_spatial_hashing_parallel(pts, S.Aabb3d(g["grid_min"], g["grid_max"]), h, context=gpu_ctx)
        ptr, idx = nl.csr
        assert np.array_equal(ptr.astype(np.int64), g["row_ptr"]) and np.array_equal(idx.astype(np.int64), g["neighbors"].astype(np.int64))
        lists = nl.get_neighborhood_lists()
        assert len(lists) == pts.shape[0] and all(i not in set(l.tolist()) for i, l in list(enumerate(lists))[:50])
    from splashsurf_amd.api import SplashsurfError
    with pytest.raises(SplashsurfError):  # particle outside the domain: the reference panics
        S.neighborhood_search_spatial_hashing_parallel(pts, S.Aabb3d(g["grid_min"], g["grid_min"] + 0.1), h, context=gpu_ctx)


def test_u64_triangles_cross_pcie_as_u32(gpu_ctx):
    """ss_result_triangles hands out the reference's index type ([usize; 3]).  For large meshes the indices cross PCIe as u32 in
    chunks and host threads widen them while the next chunk is in flight; SS_OPTION_WIDEN_ON_DEVICE = 1 widens on the device and
    copies 8 bytes per index.  Both give the same array, equal to the widened u32 accessor."""
    import ctypes as C
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context
    pts = W.tank_particles(0.35)
    kw = dict(particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, subdomain_grid=True, subdomain_grid_auto_disable=False)
    a = S.reconstruct_surface(pts, context=gpu_ctx, **kw)
    assert a.counts()[1] * 3 >= (1 << 20)  # large enough for the chunked path
    t64 = a.mesh.triangles
    assert t64.dtype == np.uint64 and np.array_equal(t64, a.mesh.triangles_u32.astype(np.uint64))
    ctx_dev = Context(0)
    ctx_dev._lib.ss_context_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert ctx_dev._lib.ss_context_set_option(ctx_dev._h, 3, 1) == 0  # SS_OPTION_WIDEN_ON_DEVICE
    b = S.reconstruct_surface(pts, context=ctx_dev, **kw)
    assert np.array_equal(b.mesh.triangles, t64)
    ctx_dev.close()


def test_host_waits_are_counted(gpu_ctx):
    """ss_stats.n_host_waits: the blocking points of a call (counts polled from mail slots, the final drain).  A plain subdomain-grid call on
    device-resident particles has seven (bounding box; copies + occupied subdomains; active blocks; over-dense blocks; MC blocks; totals;
    drain), one more per over-dense arena, list regrowth or first use of a new h."""
    import torch
    g = load_golden("config5_hilbert")
    pts, prm = golden_input(g), golden_params(g)
    d = torch.from_numpy(np.ascontiguousarray(pts)).to(device_name())
    run_gpu(gpu_ctx, d, prm)  # (sizes the lists, verifies the division for this h)
    res = run_gpu(gpu_ctx, d, prm)
    assert 7 <= res.stats["n_host_waits"] <= 8, res.stats["n_host_waits"]


def test_split_mc_offsets_gives_the_same_mesh(gpu_ctx):
    """SS_OPTION_SPLIT_MC_OFFSETS: the vertex / triangle offsets of the marching-cubes blocks from two 64-bit prefix sums (the form jobs with more
    than 838 860 surface blocks take) instead of one packed 31 + 31 bit sum: same mesh, bit for bit."""
    import ctypes as C
    import splashsurf_amd as S
    from splashsurf_amd.api import Context
    g = load_golden("config5_hilbert")
    pts, prm = golden_input(g), golden_params(g)
    a = run_gpu(gpu_ctx, pts, prm)
    ctx2 = Context(0)
    ctx2._lib.ss_context_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    assert ctx2._lib.ss_context_set_option(ctx2._h, 4, 1) == 0  # SS_OPTION_SPLIT_MC_OFFSETS
    b = run_gpu(ctx2, pts, prm)
    assert a.counts() == b.counts() and a.counts()[0] > 100000
    assert np.array_equal(a.mesh.vertices.view(np.uint32), b.mesh.vertices.view(np.uint32))
    assert np.array_equal(a.mesh.triangles_u32, b.mesh.triangles_u32) and np.array_equal(a.vertex_keys, b.vertex_keys)
    ctx2.close()


@pytest.mark.gpu
def test_hbm_bandwidth_probe_reports_plausible_rates():
    """ss_measure_hbm_bandwidth (the denominator of roofline.frac_of_measured_peak): a float4 read stream and a float4 copy over
    buffers that do not fit any cache; both rates are positive, below the data-sheet peak, and a copy (read + write bytes) is not
    slower than half the read stream."""
    from splashsurf_amd.api import Context
    ctx = Context(0)
    read_gbs, copy_gbs = ctx.measure_hbm_bandwidth(nbytes=512 << 20, repetitions=3)
    assert 500.0 < read_gbs < 8000.0, read_gbs
    assert 500.0 < copy_gbs < 8000.0, copy_gbs
    assert copy_gbs > 0.4 * read_gbs
    ctx.close()


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
@pytest.mark.parametrize("dt,grid", [(np.float32, True), (np.float32, False), (np.float64, True)])
def test_non_finite_coordinates_are_refused(gpu_ctx, bad, dt, grid):
    """A NaN or infinite coordinate has no cell: the call fails with SS_ERR_INVALID_ARGUMENT instead of indexing out of its tables (found with the
    AddressSanitizer build of tests/emu: a wild store in k_sorted_gather_runs).  The reference's f32::min / max skip the NaN in the AABB and its
    `NaN as i64` files the particle under cell 0; nothing useful follows from that either.  With an explicit particle AABB the particle is outside
    it and filtered out like any other (lib.rs:369-406): the result equals the one without the particle."""
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import SplashsurfError
    good = W.tank_particles(0.06).astype(dt)
    pts = good.copy()
    pts[17, 1] = bad
    kw = dict(particle_radius=0.005, smoothing_length=2.0, cube_size=0.5 if grid else 1.0, subdomain_grid=grid, subdomain_grid_auto_disable=False, simd=False, context=gpu_ctx)
    with pytest.raises(SplashsurfError) as e:
        S.reconstruct_surface(pts, **kw)
    assert "finite" in str(e.value)
    ok = S.reconstruct_surface(good, **kw)  # the context is fine afterwards
    assert ok.mesh.vertices.shape[0] > 0
    if grid and dt == np.float32:
        lo, hi = good.min(axis=0) - 1.0, good.max(axis=0) + 1.0
        box = dict(aabb_min=[float(x) for x in lo], aabb_max=[float(x) for x in hi])
        a = S.reconstruct_surface(pts, **kw, **box)
        b = S.reconstruct_surface(np.delete(good, 17, axis=0), **kw, **box)
        assert np.array_equal(a.mesh.vertices.view(np.uint32), b.mesh.vertices.view(np.uint32)) and np.array_equal(a.mesh.triangles_u32, b.mesh.triangles_u32)
