"""GPU tests of the NATIVE multi-GPU path (-m gpu): ss_dist_reconstruct_* / ss_dist_assemble (csrc/ss_dist.hip), the whole
sharded reconstruction behind the C ABI.

A 1-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so the algorithm is exercised with the library's
in-process transport: k host threads, one HIP context and one communicator each, all on GPU 0 -- brick partition, the three
sparse all-to-alls (positions, halo densities, shared-vertex ids), both phases of the engine and the rank-owned mesh assembly
run exactly as they do over RCCL; only the byte transport differs (device-to-device copies instead of ncclSend/ncclRecv).
The merged result must equal the single-context reconstruction bit for bit.  A one-rank RCCL communicator additionally drives
the real RCCL library (dlopen, ncclCommInitRank, ncclAllGather / ncclAllReduce on the context's stream)."""
import os
import threading

import numpy as np
import pytest

import mesh_compare as MC

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(__file__), "data")


def _case(name):
    from splashsurf_amd import workloads as W
    if name == "dam_break_n16":
        return np.load(os.path.join(DATA, "double_dam_break_frame_26_4732_particles.npy")), 0.025, 2.0, 1.1, 16
    if name == "hilbert_n32":
        return np.load(os.path.join(DATA, "hilbert_46843_particles.npy"))[::3].copy(), 0.025, 2.0, 0.9, 32
    if name == "tank_crop":
        return W.tank_particles(0.2), 0.005, 2.0, 0.5, 64
    raise KeyError(name)


def _params(r, l, c, n_cubes, dt, simd):
    from splashsurf_amd.api import Parameters
    return Parameters(particle_radius=r, compact_support_radius=dt(2.0 * l * r), cube_size=dt(c * r), subdomain_num_cubes_per_dim=n_cubes, auto_disable=False,
                      enable_simd=simd)


def _run_ranks(pts, prm, world, transport="local", take_turns=False, feedback=False, steps=2):
    """Returns per-rank dicts (info, partition, gids, rho, mesh piece).  Rank r contributes the r-th contiguous slice of pts."""
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context
    ctxs = [Context(0) for _ in range(world)]
    comms = D.NativeComm.local_group(ctxs, take_turns=take_turns) if transport == "local" else [D.NativeComm.rccl(ctxs[0], rank=0, world=1)]
    cut = [int(round(pts.shape[0] * k / world)) for k in range(world + 1)]
    out, errors = [None] * world, []

    def worker(q):
        try:
            if feedback:
                comms[q].set_balance_feedback(True)
            sh = D.NativeSharded(comms[q], prm)
            for _ in range(steps):  # the second step reuses every buffer
                res = sh.step(np.ascontiguousarray(pts[cut[q]:cut[q + 1]]))
                info = sh.assemble()
            out[q] = dict(info=info, partition=sh.partition(), gids=sh.global_ids(), rho=res.particle_densities.copy(), piece=sh.mesh_piece(),
                          local_counts=res.counts(), stats=res.stats)
            sh.result._free()
        except Exception as e:  # a failing rank must not leave the others waiting for the timeout silently
            errors.append((q, repr(e)))

    th = [threading.Thread(target=worker, args=(q,)) for q in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.close()
    assert not errors, errors
    return out


def _check_against_direct(pts, prm, ranks, gpu_ctx, expect_shared=True):
    direct = gpu_ctx.reconstruct(pts, prm)
    U = np.uint32 if pts.dtype == np.float32 else np.uint64
    world = len(ranks)
    # partition: identical on every rank, bricks tile the subdomain grid
    for r in ranks[1:]:
        assert r["partition"]["bricks"] == ranks[0]["partition"]["bricks"]
    ns = direct.subdomain_grid.ncells_per_dim
    cover = np.zeros(tuple(int(x) for x in ns), np.int32)
    for lo, hi in ranks[0]["partition"]["bricks"]:
        cover[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] += 1
    assert (cover == 1).all()
    assert sum(ranks[0]["partition"]["owned"]) == pts.shape[0]
    # densities: every particle is owned by exactly one rank; owned values and received halo values equal the direct run
    rho_ref = direct.particle_densities
    seen = np.zeros(pts.shape[0], np.int32)
    for q, r in enumerate(ranks):
        g = r["gids"].astype(np.int64)
        assert np.all(np.diff(g) > 0)  # ascending global ids
        assert np.array_equal(r["rho"].view(U), rho_ref[g].view(U)), "rank %d: densities of held particles differ" % q
        assert r["info"]["n_held"] == g.size and r["info"]["n_total"] == pts.shape[0]
    assert sum(r["info"]["n_owned"] for r in ranks) == pts.shape[0]
    # mesh: concatenation over ranks of (owned vertices, triangles with global ids)
    V = np.concatenate([r["piece"][0] for r in ranks])
    K = np.concatenate([r["piece"][1] for r in ranks])
    T = np.concatenate([r["piece"][2] for r in ranks])
    voff = 0
    for r in ranks:
        assert r["info"]["vertex_offset"] == voff
        voff += r["info"]["n_vertices_owned"]
        assert r["info"]["n_vertices_total"] == V.shape[0] and r["info"]["n_triangles_total"] == T.shape[0]
    assert np.unique(K).size == K.size  # every vertex exists once: nothing to de-duplicate
    assert V.shape[0] == direct.counts()[0] and T.shape[0] == direct.counts()[1]
    cmp = MC.compare_keyed(V, K, T, direct.mesh.vertices, direct.vertex_keys, direct.mesh.triangles)
    assert cmp["keys_equal"] and cmp["triangles_equal"] and cmp["vertices_bit_equal"], cmp
    if world > 1:
        assert any(r["info"]["bytes_sent_positions"] > 0 for r in ranks)
        if expect_shared:  # the surface crosses brick faces: shared vertices are emitted by every holder and resolved through the owner
            assert sum(r["local_counts"][0] for r in ranks) > V.shape[0]
            assert any(r["info"]["bytes_sent_assembly"] > 0 for r in ranks)
    return direct


@pytest.mark.parametrize("case,world,dt,simd", [("dam_break_n16", 2, np.float32, 0), ("dam_break_n16", 3, np.float32, 1), ("hilbert_n32", 4, np.float32, 0),
                                                ("hilbert_n32", 8, np.float32, 1), ("dam_break_n16", 5, np.float64, 0), ("tank_crop", 8, np.float32, 1)])
def test_native_ranks_reproduce_single_context(gpu_ctx, case, world, dt, simd):
    pts, r, l, c, n_cubes = _case(case)
    pts = np.ascontiguousarray(pts, dtype=dt)
    prm = _params(r, l, c, n_cubes, dt, simd)
    ranks = _run_ranks(pts, prm, world)
    _check_against_direct(pts, prm, ranks, gpu_ctx)
    if case == "tank_crop":  # 18 subdomains for 8 ranks: whole-subdomain bricks cannot balance better than this
        assert ranks[0]["partition"]["imbalance_owned"] <= 1.6, ranks[0]["partition"]


def test_native_full_s40m_tank_four_ranks(gpu_ctx):
    """BASELINE config 4 at FULL size through the native multi-GPU path: the 39.8 M-particle S40M-tank cut into four bricks (in-process
    transport, the ranks taking turns on the device like bench.py --pseudo-ranks), exchanges and rank-owned assembly included; densities of
    every held particle, vertices, edge keys and triangles of the merged mesh equal the single-context reconstruction bit for bit, and that
    reconstruction equals the reference wheel's own mesh of this input (vertex-id multiset, triangle set, all 39.8 M densities)."""
    from splashsurf_amd import workloads as W
    wl = W.WORKLOADS["s40m_tank"]
    pts = wl["gen"]()
    assert pts.shape[0] > 39_000_000
    prm = _params(wl["particle_radius"], wl["smoothing_length"], wl["cube_size"], 64, np.float32, 0)  # the scalar arithmetic: the mode pinned to the wheel
    ranks = _run_ranks(pts, prm, 4, take_turns=True)
    direct = _check_against_direct(pts, prm, ranks, gpu_ctx)
    # ... and that single-context mesh IS the reference wheel's (tests/golden/config4_s40m_tank.npz): merged == direct == wheel
    from conftest import load_golden
    from test_gpu_fullsize import assert_equals_the_wheels_digest
    assert_equals_the_wheels_digest(direct, load_golden("config4_s40m_tank"), False, "config4_s40m_tank (merged mesh of four ranks == this)")
    part = ranks[0]["partition"]
    assert part["imbalance_owned"] <= 1.05, part
    for r in ranks:  # the phases have their own clocks (round 2 billed phase 1 to the density exchange) and the turn timer ran
        i = r["info"]
        assert i["ms_phase1"] > 0.0 and i["ms_phase2"] > 0.0
        assert 0.0 < i["ms_own_turns"] < i["ms_partition"] + i["ms_position_exchange"] + i["ms_phase1"] + i["ms_density_exchange"] + i["ms_phase2"] + i["ms_assembly"]


def test_native_partition_feedback_keeps_the_mesh(gpu_ctx):
    """ss_comm_set_balance_feedback: from the second step on the bricks are balanced by the cost every rank measured in the previous step (with
    hysteresis).  Whatever partition comes out, it is the same on every rank, tiles the subdomain grid, and the merged mesh and densities equal the
    single-context reconstruction bit for bit; the step counts its ten communication steps."""
    pts, r, l, c, n_cubes = _case("tank_crop")
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    prm = _params(r, l, c, 16, np.float32, 0)  # 16-cell subdomains: enough of them for the weights to move a plane
    ranks = _run_ranks(pts, prm, 4, take_turns=True, feedback=True, steps=5)
    _check_against_direct(pts, prm, ranks, gpu_ctx)
    for r_ in ranks:
        assert r_["info"]["n_collectives"] == 10
        assert r_["info"]["ms_device"] > 0.0


def test_native_more_ranks_than_subdomains(gpu_ctx):
    """Six ranks, a 1x1x2 subdomain grid: four ranks get empty bricks and still take part in every collective."""
    pts = np.load(os.path.join(DATA, "cube_2366_particles.npy")).astype(np.float32)
    prm = _params(0.025, 2.0, 0.75, 64, np.float32, 0)
    ranks = _run_ranks(pts, prm, 6)
    direct = _check_against_direct(pts, prm, ranks, gpu_ctx, expect_shared=False)
    n_sub = int(np.prod(direct.subdomain_grid.ncells_per_dim))
    assert sum(1 for r in ranks if r["info"]["n_owned"] == 0) >= 6 - n_sub


def test_native_one_rank_rccl(gpu_ctx):
    """The RCCL transport itself with a one-rank communicator: dlopen of librccl, ncclGetUniqueId / ncclCommInitRank, the small
    collectives on the context's stream.  (Two ranks need two GPUs.)"""
    pts, r, l, c, n_cubes = _case("dam_break_n16")
    pts = pts.astype(np.float32)
    prm = _params(r, l, c, n_cubes, np.float32, 1)
    ranks = _run_ranks(pts, prm, 1, transport="rccl")
    _check_against_direct(pts, prm, ranks, gpu_ctx)


def test_native_missing_peer_times_out_loudly(monkeypatch):
    """A rank that never calls in makes the others fail with an error after SPLASH_COMM_TIMEOUT_S instead of hanging."""
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, SplashsurfError
    monkeypatch.setenv("SPLASH_COMM_TIMEOUT_S", "2")
    pts, r, l, c, n_cubes = _case("dam_break_n16")
    ctxs = [Context(0), Context(0)]
    comms = D.NativeComm.local_group(ctxs)
    sh = D.NativeSharded(comms[0], _params(r, l, c, n_cubes, np.float32, 0))
    with pytest.raises(SplashsurfError):
        sh.step(pts[:2000].astype(np.float32))  # rank 1 never shows up
    sh.result._free()
    for cm in comms:
        cm.destroy()
    for cx in ctxs:
        cx.close()


def test_native_ranks_that_disagree_on_the_feedback_setting_fail_loudly(monkeypatch):
    """ss_comm_set_balance_feedback on ONE of two ranks only: the ranks would cut different bricks from the second step on.  The digest of the
    feedback state that travels with every step's first all-gather makes both fail with an error instead."""
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, SplashsurfError
    monkeypatch.setenv("SPLASH_COMM_TIMEOUT_S", "20")
    pts, r, l, c, n_cubes = _case("dam_break_n16")
    pts = pts.astype(np.float32)
    prm = _params(r, l, c, n_cubes, np.float32, 0)
    ctxs = [Context(0), Context(0)]
    comms = D.NativeComm.local_group(ctxs)
    comms[1].set_balance_feedback(True)
    cut = [0, pts.shape[0] // 2, pts.shape[0]]
    raised = [None, None]

    def worker(q):
        sh = D.NativeSharded(comms[q], prm)
        try:
            sh.step(np.ascontiguousarray(pts[cut[q]:cut[q + 1]]))
        except SplashsurfError as e:
            raised[q] = str(e)
        sh.result._free()

    th = [threading.Thread(target=worker, args=(q,)) for q in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c_ in comms:
        c_.destroy()
    for c_ in ctxs:
        c_.close()
    assert raised[0] and raised[1] and "feedback" in raised[0] and "feedback" in raised[1], raised


def test_native_brick_resident_time_series_ships_halos_only(gpu_ctx):
    """A time series that keeps its particles where they are owned: after a first frame the cloud is re-dealt by owner brick
    (distributed.brick_owner_of on the partition that frame produced), every rank hands in the particles of its OWN brick.  The merged mesh equals
    the single-context reconstruction of the re-dealt cloud bit for bit (the input order is part of the input, as for the reference), and the
    position exchange carries halos instead of whole slices (on this small cloud with 16-cell subdomains the ghost layers are two thirds of a brick, so
    the bytes only fall by a third; at the size of BASELINE config 4 the halos are 2.6 % of the owned particles,
    profiles/r06_s40m_tank_pseudo_ranks_8.json: 128 k of 4.9 M per rank)."""
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context
    pts, r, l, c, n_cubes = _case("tank_crop")
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    prm = _params(r, l, c, 16, np.float32, 0)
    world = 4
    ctxs = [Context(0) for _ in range(world)]
    comms = D.NativeComm.local_group(ctxs)
    cut = [int(round(pts.shape[0] * k / world)) for k in range(world + 1)]
    shards = [D.NativeSharded(comms[q], prm) for q in range(world)]
    first, second, out, errors = [None] * world, [None] * world, [None] * world, []
    bar = threading.Barrier(world)
    dealt = {}

    def worker(q):
        try:
            sh = shards[q]
            res = sh.step(np.ascontiguousarray(pts[cut[q]:cut[q + 1]]))
            first[q] = sh.assemble()
            if q == 0:  # one rank deals for all (every rank sees the same partition and subdomain grid)
                owner = D.brick_owner_of(pts, res.subdomain_grid, sh.partition()["bricks"])
                assert owner.min() >= 0
                dealt["parts"] = [np.ascontiguousarray(pts[owner == k]) for k in range(world)]
            bar.wait()
            for _ in range(2):
                res = sh.step(dealt["parts"][q])
                second[q] = sh.assemble()
            out[q] = dict(info=second[q], partition=sh.partition(), gids=sh.global_ids(), rho=res.particle_densities.copy(), piece=sh.mesh_piece(),
                          local_counts=res.counts(), stats=res.stats)
            sh.result._free()
        except Exception as e:
            errors.append((q, repr(e)))
            try:
                bar.abort()
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=(q,)) for q in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c_ in comms:
        c_.destroy()
    for c_ in ctxs:
        c_.close()
    assert not errors, errors
    redealt = np.ascontiguousarray(np.concatenate(dealt["parts"]))
    assert redealt.shape == pts.shape
    _check_against_direct(redealt, prm, out, gpu_ctx, expect_shared=False)  # (four fluid blocks of this crop: no surface through a brick face)
    before = sum(i["bytes_sent_positions"] for i in first)
    after = sum(i["bytes_sent_positions"] for i in second)
    assert before > 0 and after * 5 <= before * 4, (before, after)
    # (the bisection breaks ties by how far the ranks' inputs extend along each axis, so the partition of the re-dealt frames may differ from the one the
    # deal was made for: some particles then still travel -- bytes, never correctness)


def test_native_non_finite_input_on_one_rank_fails_on_every_rank(monkeypatch):
    """One rank's share holds a NaN: the flag travels with the step's first all-gather and EVERY rank refuses the step (nobody waits for a peer that gave up)."""
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context, SplashsurfError
    monkeypatch.setenv("SPLASH_COMM_TIMEOUT_S", "20")
    pts, r, l, c, n_cubes = _case("dam_break_n16")
    pts = pts.astype(np.float32)
    prm = _params(r, l, c, n_cubes, np.float32, 0)
    ctxs = [Context(0), Context(0)]
    comms = D.NativeComm.local_group(ctxs)
    cut = [0, pts.shape[0] // 2, pts.shape[0]]
    shares = [np.ascontiguousarray(pts[cut[q]:cut[q + 1]]) for q in range(2)]
    shares[1][5, 2] = np.nan
    raised = [None, None]

    def worker(q):
        sh = D.NativeSharded(comms[q], prm)
        try:
            sh.step(shares[q])
        except SplashsurfError as e:
            raised[q] = str(e)
        sh.result._free()

    th = [threading.Thread(target=worker, args=(q,)) for q in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c_ in comms:
        c_.destroy()
    for c_ in ctxs:
        c_.close()
    assert raised[0] and raised[1] and "finite" in raised[0] and "finite" in raised[1], raised
