"""One rank of the RCCL parity test (tests/test_gpu_dist_rccl.py), started by torch.distributed.run.

Every rank binds GPU LOCAL_RANK, creates a library context and an RCCL communicator (`ss_comm_create_rccl`; the unique id
travels over a gloo group), contributes the RANK-th contiguous slice of the case's particle cloud to
`ss_dist_reconstruct_f32/_f64` + `ss_dist_assemble` (grouped ncclSend/ncclRecv between the GPUs: positions, halo densities,
shared vertex ids) and writes what it holds -- global ids, densities, its owned vertices / edge keys / triangles, the
partition and the exchange statistics -- to <outdir>/rank<r>.pkl.  The test process compares the concatenation over ranks
with a single-context reconstruction bit for bit (the reference's unit of parallelism is the subdomain,
dense_subdomains.rs:521-526, its join of shared vertices the stitching pass, :1693-1733).

usage: python -m torch.distributed.run ... tests/rccl_rank_worker.py <case> <dtype> <simd> <outdir>
"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def case_particles(name):
    from splashsurf_amd import workloads as W
    data = os.path.join(ROOT, "tests", "data")
    if name == "dam_break_n16":
        return np.load(os.path.join(data, "double_dam_break_frame_26_4732_particles.npy")), 0.025, 2.0, 1.1, 16
    if name == "hilbert_n32":
        return np.load(os.path.join(data, "hilbert_46843_particles.npy"))[::3].copy(), 0.025, 2.0, 0.9, 32
    if name == "tank_crop":
        # the 1.24 M-particle crop of S40M-tank's geometry (BASELINE config 4), in a seeded random order: every rank's contiguous share
        # is spread over the whole domain, so every ordered pair of GPUs exchanges particles, densities and shared vertices
        pts = W.tank_particles(0.5)
        return pts[np.random.default_rng(7).permutation(pts.shape[0])].copy(), 0.005, 2.0, 0.5, 64
    raise KeyError(name)


def expect_shared_vertices(name, world):
    """Does the surface cross brick faces (then shared vertices exist and the assembly exchange carries bytes)?  Checked on one GPU
    with the in-process transport by test_local_transport_twin."""
    if world < 2:
        return False
    if name == "tank_crop":
        return world >= 4  # two ranks: one fluid block each, the blocks do not touch
    return True


def case_params(r, l, c, n_cubes, dt, simd):
    from splashsurf_amd.api import Parameters
    return Parameters(particle_radius=r, compact_support_radius=dt(2.0 * l * r), cube_size=dt(c * r), subdomain_num_cubes_per_dim=n_cubes, auto_disable=False,
                      enable_simd=simd)


def main():
    case, dtype, simd, outdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    # no-GPU container (tests/test_emu_kernels.py): the emulated library (SPLASHSURF_HIP_LIB) with the stand-in RCCL (SPLASH_RCCL_LIB), every
    # rank on the emulator's one "device", host arrays as input -- the RCCL branch of ss_dist.hip between real processes
    emulated = os.environ.get("SPLASH_EMULATED_RANKS") == "1"
    if emulated:
        local = 0
    else:
        torch.cuda.set_device(local)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from splashsurf_amd import distributed as D
    from splashsurf_amd.api import Context
    dt = np.float64 if dtype == "f64" else np.float32
    pts, r, l, c, n_cubes = case_particles(case)
    pts = np.ascontiguousarray(pts, dtype=dt)
    prm = case_params(r, l, c, n_cubes, dt, simd)
    ctx = Context(local)
    comm = D.NativeComm.rccl(ctx)
    assert comm.kind == "rccl" and comm.world == world
    cut = [int(round(pts.shape[0] * k / world)) for k in range(world + 1)]
    sh = D.NativeSharded(comm, prm)
    mine = np.ascontiguousarray(pts[cut[rank]:cut[rank + 1]])
    if not emulated:
        mine = torch.from_numpy(mine).to("cuda:%d" % local)  # HBM-resident share, as in bench.py --gpus N
    for _ in range(2):  # the second step reuses every buffer and every RCCL connection
        res = sh.step(mine)
        info = sh.assemble()
    out = dict(info=info, partition=sh.partition(), gids=sh.global_ids(), rho=res.particle_densities.copy(), piece=sh.mesh_piece(),
               local_counts=res.counts(), stats=dict(res.stats), device=rank if emulated else local,
               device_name="hip-emu" if emulated else torch.cuda.get_device_name(local))
    with open(os.path.join(outdir, "rank%d.pkl" % rank), "wb") as f:
        pickle.dump(out, f, protocol=4)
    dist.barrier()
    sh.result._free()
    comm.destroy()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
