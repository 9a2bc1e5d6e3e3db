"""CPU tests of the multi-process path (world_size 2, gloo): splashsurf_amd/distributed.py with the
oracle plugged in as the per-rank engine must reproduce the single-process oracle bit for bit
(densities) and key for key (mesh)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class OracleEngine:
    """Per-rank engine backed by the CPU oracle (tests only)."""

    class _Mesh:
        pass

    def __init__(self, O, params):
        self.O, self.params = O, params

    def grid_for_domain(self, dmin, dmax):
        g, sg, margin = self.O.grid_for_domain(self.params, dmin, dmax)
        return g["aabb_min"], float(sg["cell_size"]), [int(x) for x in sg["n_cells"]], margin, int(self.params.subdomain_num_cubes_per_dim)

    def begin(self, local_pts, shard):
        self._pts = local_pts.cpu().numpy()
        self._shard = shard
        rho = self.O.shard_densities(self._pts, self.params, shard.domain_min, shard.domain_max, shard.sub_lo, shard.sub_hi)
        return torch.from_numpy(rho)

    def finish(self, rho):
        sh = self._shard
        r = self.O.shard_reconstruct(self._pts, rho.cpu().numpy(), self.params, sh.domain_min, sh.domain_max, sh.sub_lo, sh.sub_hi)
        res = OracleEngine._Mesh()
        res.mesh = OracleEngine._Mesh()
        res.mesh.vertices, res.mesh.triangles, res.vertex_keys = r.vertices, r.triangles, r.vertex_keys
        res.stats = {}
        res.subdomain_stats = lambda: (r.n_subdomains, r.n_subdomain_particles)
        return res


def _worker(rank, world, port, case, out_path, exchange="p2p"):
    os.environ["SPLASH_EXCHANGE"] = exchange
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from splashsurf_amd import distributed as D
        pts, r, l, c, n_cubes = _case(case)
        par = O.make_params_relative(r, l, c, subdomain_num_cubes_per_dim=n_cubes, num_threads=2)
        # contiguous split of the input: global particle order = concatenation by rank
        cut = [0] + [int(round(pts.shape[0] * (k + 1) / world)) for k in range(world)]
        if case == "tank_slabs":  # the bench's weak-scaling input: one tank per rank, stacked along y
            from splashsurf_amd import workloads as W
            cut = [0] + [int(x) for x in np.cumsum([W.tank_slab_particles(k, world, scale=0.06).shape[0] for k in range(world)])]
        sh = D.ShardedReconstruction(OracleEngine(O, par), "cpu")
        sh.load_local_particles(pts[cut[rank]:cut[rank + 1]])
        step = sh.step()
        merged = sh.gather_mesh(step)
        rho_global = sh.gather_densities()
        if rank == 0:
            v, k, t = merged
            np.savez(out_path, vertices=v, keys=k, triangles=t, rho=rho_global.numpy(),
                     slab=np.array([step.shard.sub_lo, step.shard.sub_hi]), n_local=np.int64(step.ids.shape[0]))
    finally:
        dist.destroy_process_group()


def _case(name):
    data = os.path.join(ROOT, "tests", "data")
    if name == "dam_break_n16":
        return np.load(os.path.join(data, "double_dam_break_frame_26_4732_particles.npy")), 0.025, 2.0, 1.1, 16
    if name == "hilbert_n32":
        return np.load(os.path.join(data, "hilbert_46843_particles.npy"))[::4].copy(), 0.025, 2.0, 1.0, 32
    if name == "tank_slabs":
        from splashsurf_amd import workloads as W
        return np.concatenate([W.tank_slab_particles(k, 2, scale=0.06) for k in range(2)]), 0.005, 2.0, 0.5, 64
    if name == "lattice_n8":
        return np.load(os.path.join(data, "cube_2366_particles.npy")), 0.025, 2.0, 0.75, 8
    raise KeyError(name)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case,world,exchange", [("dam_break_n16", 2, "p2p"), ("hilbert_n32", 2, "p2p"), ("lattice_n8", 2, "p2p"), ("dam_break_n16", 3, "p2p"),
                                                 ("dam_break_n16", 3, "allgather"), ("tank_slabs", 2, "p2p"),
                                                 # four and five bricks: cuts along two axes, edges shared by up to four ranks
                                                 ("dam_break_n16", 4, "p2p"), ("hilbert_n32", 5, "p2p")])
def test_ranks_reproduce_single_process(tmp_path, oracle, case, world, exchange):
    import mesh_compare as MC
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), case, out, exchange), nprocs=world, join=True)
    got = np.load(out)
    pts, r, l, c, n_cubes = _case(case)
    ref = oracle.reconstruct_surface(pts, oracle.make_params_relative(r, l, c, subdomain_num_cubes_per_dim=n_cubes))
    # the domain really was split: rank 0 did not hold all particles
    assert got["slab"][1].max() > 0 and int(got["n_local"]) < pts.shape[0]
    assert np.array_equal(got["rho"].view(np.uint32), ref.particle_densities.view(np.uint32))
    cmp = MC.compare_keyed(got["vertices"], got["keys"], got["triangles"], ref.vertices, ref.vertex_keys, ref.triangles)
    assert cmp["keys_equal"] and cmp["triangles_equal"], cmp
    # face vertices: each rank's lowest-index subdomain wins; identical to the single-process choice
    assert cmp["vertices_bit_equal"], cmp
    if case == "tank_slabs":
        # a 2x2x2 subdomain grid: exactly one axis is cut, into two bricks of one layer each
        par = oracle.make_params_relative(r, l, c, subdomain_num_cubes_per_dim=n_cubes)
        _, sg, _ = oracle.grid_for_domain(par, pts.min(axis=0), pts.max(axis=0))
        ns = [int(x) for x in sg["n_cells"]]
        hi = [int(x) for x in got["slab"][1]]
        assert ns[0] == ns[1] == ns[2] == 2, ns
        assert sorted(hi) == [1, 2, 2], (hi, ns)


def test_native_partition_rule_equals_the_host_mirror():
    """ss_dist_partition (csrc/ss_dist.hip, host-only code: runs without a GPU) cuts the same bricks as
    distributed.bricks_from_histogram on random and on structured histograms."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    from splashsurf_amd.distributed import bricks_from_histogram
    if not os.path.exists(G.LIB):
        G.build()
    L = C.CDLL(G.LIB)
    L.ss_dist_partition.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(5)
    for trial in range(40):
        ns = tuple(int(x) for x in rng.integers(1, 9, size=3))
        kind = trial % 4
        if kind == 0:
            hist = rng.integers(0, 5000, size=ns)
        elif kind == 1:
            hist = (rng.random(ns) < 0.3) * rng.integers(1, 100000, size=ns)
        elif kind == 2:
            hist = np.zeros(ns, np.int64)
        else:
            hist = np.full(ns, 777)
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        pref = tuple(round(float(x), 3) for x in rng.random(3)) if trial % 2 else (0.0, 0.0, 0.0)
        for world in (1, 2, 3, 5, 8, 13):
            want = bricks_from_histogram(hist, world, axis_pref=pref)
            out = (C.c_int64 * (6 * world))()
            st = L.ss_dist_partition(hist.ctypes.data_as(C.c_void_p), (C.c_int64 * 3)(*ns), world, (C.c_double * 3)(*pref), out)
            assert st == 0
            got = [(tuple(out[6 * q:6 * q + 3]), tuple(out[6 * q + 3:6 * q + 6])) for q in range(world)]
            assert got == [(tuple(a), tuple(b)) for a, b in want], (ns, world, pref, got, want)


def test_partition_is_balanced_and_contiguous():
    sys.path.insert(0, ROOT)
    from splashsurf_amd.distributed import partition_slabs
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random(100000).astype(np.float32) * 10.0)
    slabs = partition_slabs(x, 0.0, 1.0, 10, 4)
    assert slabs[0][0] == 0 and slabs[-1][1] == 10
    assert all(slabs[i][1] == slabs[i + 1][0] for i in range(3))
    sizes = [((x >= lo) & (x < hi)).sum().item() for lo, hi in slabs]
    assert max(sizes) < 1.5 * min(sizes)
    # more ranks than subdomains: trailing ranks get empty slabs, nothing is lost
    slabs = partition_slabs(x, 0.0, 5.0, 2, 4)
    assert slabs[-1][1] == 2 and sum(hi - lo for lo, hi in slabs) == 2


def _brick_cells(b):
    (a0, a1, a2), (b0, b1, b2) = b
    return max(b0 - a0, 0) * max(b1 - a1, 0) * max(b2 - a2, 0)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8, 16])
def test_bricks_tile_the_grid_and_balance(world):
    """Recursive bisection of the subdomain grid: bricks are disjoint, cover every subdomain, and split the particles of
    an S40M-tank-shaped histogram (two separated fluid blocks, whole-subdomain cut planes) within 15 % of the mean."""
    sys.path.insert(0, ROOT)
    from splashsurf_amd.distributed import bricks_from_histogram
    ns = (40, 21, 40)
    hist = np.zeros(ns, dtype=np.int64)
    hist[0:13, 0:20, 0:20] = 4800
    hist[12, :, :] //= 2       # partially filled boundary layers
    hist[0:13, 19, 0:20] //= 3
    hist[27:40, 0:20, 20:40] = 4800
    hist[27, :, :] //= 4
    bricks = bricks_from_histogram(hist, world, axis_pref=(0.1, 1.0, 1.0))
    assert len(bricks) == world
    cover = np.zeros(ns, dtype=np.int32)
    own = []
    for a, b in bricks:
        cover[a[0]:b[0], a[1]:b[1], a[2]:b[2]] += 1
        own.append(int(hist[a[0]:b[0], a[1]:b[1], a[2]:b[2]].sum()))
    assert (cover == 1).all()
    assert sum(own) == int(hist.sum())
    if world <= 8:
        assert max(own) <= 1.15 * (sum(own) / world), own
    # more ranks than subdomains: the surplus ranks get empty bricks, nothing is lost or duplicated
    tiny = bricks_from_histogram(np.ones((1, 2, 1)), 5)
    assert sum(_brick_cells(b) for b in tiny) == 2 and sum(1 for b in tiny if _brick_cells(b) == 0) == 3
    # an empty domain still yields a valid tiling
    empty = bricks_from_histogram(np.zeros((3, 3, 3)), 4)
    assert sum(_brick_cells(b) for b in empty) == 27


# ---------------------------------------------------------------------------------------------------------------
# GPU: two real processes, both on GPU 0, exchanging halos over gloo, each driving the HIP shard ABI
# (ss_shard_begin_f32 / ss_shard_finish) -- everything of the N > 1 bench path except RCCL's own transport,
# which needs a second GPU (RCCL refuses two ranks on one device).
# ---------------------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, case, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from splashsurf_amd import distributed as D
        from splashsurf_amd.api import Context, Parameters
        pts, r, l, c, n_cubes = _case(case)
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
            dev = torch.device("cuda", 0)
        else:  # the library under test is the CPU execution model of tests/emu (SPLASHSURF_HIP_LIB): its device memory is the host's
            dev = torch.device("cpu")
        prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * l * r), cube_size=np.float32(c * r),
                         subdomain_num_cubes_per_dim=n_cubes, auto_disable=False, enable_simd=False)
        cut = [0] + [int(round(pts.shape[0] * (k + 1) / world)) for k in range(world)]
        sh = D.ShardedReconstruction(D.HipEngine(Context(0), prm), dev)
        sh.load_local_particles(pts[cut[rank]:cut[rank + 1]])
        for _ in range(2):  # second step reuses every buffer
            step = sh.step()
        merged = sh.gather_mesh(step)
        rho_global = sh.gather_densities()
        if rank == 0:
            v, k, t = merged
            np.savez(out_path, vertices=v, keys=k, triangles=t, rho=rho_global.cpu().numpy(),
                     slab=np.array([step.shard.sub_lo, step.shard.sub_hi]), n_local=np.int64(step.ids.shape[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("case,world", [("dam_break_n16", 2), ("hilbert_n32", 3)])
def test_gpu_ranks_on_one_device_reproduce_single_process(tmp_path, oracle, case, world):
    import mesh_compare as MC
    out = str(tmp_path / "merged_gpu.npz")
    mp.spawn(_gpu_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    got = np.load(out)
    pts, r, l, c, n_cubes = _case(case)
    ref = oracle.reconstruct_surface(pts, oracle.make_params_relative(r, l, c, subdomain_num_cubes_per_dim=n_cubes))
    assert got["slab"][1].max() > 0 and int(got["n_local"]) < pts.shape[0]
    assert np.array_equal(got["rho"].view(np.uint32), ref.particle_densities.view(np.uint32))
    cmp = MC.compare_keyed(got["vertices"], got["keys"], got["triangles"], ref.vertices, ref.vertex_keys, ref.triangles)
    assert cmp["keys_equal"] and cmp["triangles_equal"] and cmp["vertices_bit_equal"], cmp


def test_brick_owner_of_follows_the_owner_histogram_rule():
    """distributed.brick_owner_of (the dealing rule of bench.py's brick-resident shares and of a time series that keeps its particles where they are owned):
    floor of the coordinate in units of the subdomain edge, clamped into the grid, looked up in the bricks -- every particle gets exactly one owner."""
    import types
    from splashsurf_amd import distributed as D
    grid = types.SimpleNamespace(aabb=types.SimpleNamespace(min=np.array([-1.0, 0.0, 0.0], np.float32)), cell_size=np.float32(0.5), ncells_per_dim=np.array([4, 2, 1]))
    bricks = [[[0, 0, 0], [1, 2, 1]], [[1, 0, 0], [4, 1, 1]], [[1, 1, 0], [4, 2, 1]]]
    pts = np.array([[-0.9, 0.1, 0.2], [-0.5, 0.1, 0.2], [-0.51, 0.9, 0.4], [0.99, 0.49, 0.0], [0.2, 0.5, 0.1],
                    [-7.0, -3.0, -1.0], [9.0, 9.0, 9.0], [9.0, 0.2, 0.3]], np.float32)  # (the last three lie outside the grid: clamped)
    assert D.brick_owner_of(pts, grid, bricks).tolist() == [0, 1, 0, 1, 2, 0, 2, 1]
    rng = np.random.default_rng(3)
    cloud = rng.uniform(-1.5, 1.5, size=(5000, 3)).astype(np.float32)
    owner = D.brick_owner_of(cloud, grid, bricks)
    assert owner.min() >= 0 and owner.max() <= 2 and np.bincount(owner, minlength=3).sum() == 5000
