#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X surface-reconstruction path.

    python bench.py --gpus N --steps K --warmup W [--workload s10m_tank|s1m|s10m_cube|s40m_tank|tank_small]

A "step" is one full pass of the hot path (ss_reconstruct_surface_f32: binning, densities, level-set
splat, marching cubes, global numbering) over one batch of synthetic particles that is ALREADY
RESIDENT IN HBM when the timed region starts; the mesh stays in HBM (counts are read back).
Metric (BASELINE.json): Mparticles/s end-to-end reconstruct; plus the splat kernel's achieved
algorithmic HBM GB/s against the 8 TB/s peak ("roofline") and the CPU oracle timed on the host cores
on a bounded sample of the same workload ("cpu_baseline", a reported baseline only).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the global domain is sharded into
slabs of subdomains along y, rank r reconstructs the surface of slab r (weak scaling: every rank
brings its own tank of particles); per-particle densities of halo particles are exchanged with one
RCCL all-gather (splashsurf_amd/distributed.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="s10m_tank")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed steps of the named workload (no host-input variants, no other configs, no CPU baseline): "
                         "the command to profile, so that rocprofv3's per-kernel averages are those of this workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU (sharded) code path even with one rank")
    ap.add_argument("--cpu-sample-scale", type=float, default=0.5, help="tank scale of the CPU-baseline sample (0.5 => 1.25M particles)")
    return ap.parse_args()


def cpu_baseline(workload, scale):
    """Time the CPU oracle (port of the reference's scalar path) on a bounded sample of the workload."""
    from oracle import oracle as O
    from splashsurf_amd import workloads as W
    wl = W.WORKLOADS[workload]
    if workload in ("s10m_tank", "tank_small"):
        pts = W.tank_particles(scale if workload == "s10m_tank" else 0.08)
        sample = "tank_particles(scale=%g): %d particles, same r/l/c as the workload" % (scale if workload == "s10m_tank" else 0.08, pts.shape[0])
    elif workload == "s1m":
        pts = wl["gen"]()[:250_000] * np.float32(0.63)  # same number density, 1/4 of the particles
        sample = "first 250k particles of S1M scaled to keep the number density"
    else:
        pts = W.uniform_cube_particles(1_000_000, 12346) * np.float32(0.464)
        sample = "1M uniform particles at the number density of S10M-cube"
    par = O.make_params_relative(wl["particle_radius"], wl["smoothing_length"], wl["cube_size"])
    t0 = time.perf_counter()
    res = O.reconstruct_surface(pts, par)
    dt = time.perf_counter() - t0
    return {
        "value": round(pts.shape[0] / dt / 1e6, 4), "unit": "Mparticles/s", "cores": res.threads_used, "kind": "port",
        "sample": sample + "; %.2f s wall, %d vertices / %d triangles" % (dt, res.vertices.shape[0], res.triangles.shape[0]),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import splashsurf_amd as S
    from splashsurf_amd import workloads as W
    from splashsurf_amd.api import Context, Parameters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded_path = world > 1 or args.force_sharded
    if sharded_path and "RANK" in os.environ:
        dist.init_process_group(backend="nccl", device_id=dev)

    wl = W.WORKLOADS[args.workload]
    r = wl["particle_radius"]
    prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r),
                     cube_size=np.float32(wl["cube_size"] * r), auto_disable=False)
    ctx = Context(local_rank)

    if not sharded_path:
        pts = wl["gen"]()
        n_total = pts.shape[0]
        d_pts = torch.from_numpy(pts).to(dev)
        torch.cuda.synchronize()
        out = None

        def step():
            nonlocal out
            out = ctx.reconstruct(d_pts, prm, out=out)
            return out
    else:
        from splashsurf_amd import distributed as D
        if args.workload not in ("s10m_tank", "tank_small"):
            raise SystemExit("multi-GPU bench supports the tank workloads")
        scale = 1.0 if args.workload == "s10m_tank" else 0.08
        pts = W.tank_slab_particles(rank, world, scale=scale, particle_radius=r)
        n_total = pts.shape[0] * world
        sharded = D.ShardedReconstruction(D.HipEngine(ctx, prm), dev)
        sharded.load_local_particles(pts)

        def step():
            return sharded.step()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if sharded_path:
        sharded.timings = {}
    barrier()
    t0 = time.perf_counter()
    k3_ms = []
    last = None
    for _ in range(args.steps):
        last = step()
        # dominant splat kernel: k_splat_accumulate (small tiles) unless most blocks are over-dense and go through k_splat_large
        s_ = last.stats
        t_acc = s_.get("ms_levelset_accumulate", 0.0)
        t_large = s_["ms_levelset"] - s_.get("ms_levelset_gather", 0.0) - t_acc  # compaction of the overflow queue + k_splat_large
        k3_ms.append((t_acc, t_large))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    st = last.stats
    n_occ, n_subp = last.subdomain_stats()
    nsc = int(prm.subdomain_num_cubes_per_dim) + 1
    # algorithmic bytes of the splat: 16 B per subdomain particle (x,y,z,rho incl. ghosts)
    # + 4 B per level-set value of every occupied subdomain ((n+1)^3 points each)
    alg_bytes = 16.0 * n_subp + 4.0 * n_occ * nsc ** 3
    k3_acc, k3_large = (float(v) for v in np.mean(np.asarray(k3_ms), axis=0))
    k3_name = "k_splat_accumulate" if k3_acc >= k3_large else "k_splat_large"
    k3 = max(k3_acc, k3_large) * 1e-3
    achieved = alg_bytes / k3 / 1e9 if k3 > 0 else 0.0
    line = {
        "metric": "Mparticles/s end-to-end reconstruct",
        "value": round(n_total * args.steps / dt / 1e6, 3),
        "unit": "Mparticles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": args.workload, "n_particles": int(n_total), "particle_radius": r, "smoothing_length": wl["smoothing_length"],
            "cube_size": wl["cube_size"], "n_vertices": int(st["n_vertices"]), "n_triangles": int(st["n_triangles"]),
            "input": "HBM-resident (x,y,z) f32", "output": "mesh in HBM", "parallelism": "1 GPU" if world == 1 else "y-slabs of subdomains x%d" % world,
        },
        "roofline": {
            "kernel": k3_name, "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 5), "traffic": None, "algorithmic_bytes": alg_bytes, "kernel_ms": round(k3 * 1e3, 4),
            "note": "algorithmic bytes = 16 B x %d subdomain particles + 4 B x %d subdomains x %d^3 points (rank 0); kernel is FP32-VALU bound at this cube radius (DESIGN.md)" % (n_subp, n_occ, nsc),
        },
        "stages_ms": {k: round(v, 4) for k, v in st.items() if k.startswith("ms_")},
        "splat_blocks": {"active": int(st.get("n_active_blocks", 0)), "large_tile": int(st.get("n_large_tile_blocks", 0))},
    }
    if not sharded_path and not args.main_only:
        # secondary figure (never `value`): the same call with HOST-resident input and the mesh copied back
        # to pinned host memory (H2D + all kernels + D2H), i.e. what a host-only caller of the C ABI sees
        try:
            host_pts = pts
            t_io = []
            for _ in range(3):
                t1 = time.perf_counter()
                r_io = ctx.reconstruct(host_pts, prm, out=out)
                _v, _t = r_io.mesh_views()  # D2H into the library's pinned host buffers, no further copy
                t_io.append(time.perf_counter() - t1)
            line["pcie_inclusive"] = {"value": round(n_total / min(t_io) / 1e6, 3), "unit": "Mparticles/s", "ms_per_step": round(min(t_io) * 1e3, 3),
                                      "note": "host (pageable numpy) input via ss_reconstruct_surface_inplace_f32, vertices + u32 triangles fetched through "
                                              "ss_result_vertices / ss_result_triangles_u32 (pinned host buffers); best of 3"}
        except Exception as e:
            line["pcie_inclusive"] = {"value": None, "note": "failed: %r" % (e,)}
        # same host-to-host call, but two frames in flight: two contexts (one HIP stream each) driven by two host threads, so
        # the H2D / D2H copies of one frame overlap the kernels of the other (a time series of frames is the real workload)
        try:
            import threading
            ctxs = [ctx, Context(local_rank)]
            outs = [out, None]
            frames = 3

            def worker(i):
                for _ in range(frames):
                    outs[i] = ctxs[i].reconstruct(host_pts, prm, out=outs[i])
                    outs[i].mesh_views()

            for i in range(2):  # warm-up of the second context's buffers
                worker_out = ctxs[i].reconstruct(host_pts, prm, out=outs[i])
                outs[i] = worker_out
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt2 = time.perf_counter() - t1
            line["pcie_pipelined"] = {"value": round(n_total * 2 * frames / dt2 / 1e6, 3), "unit": "Mparticles/s", "ms_per_frame": round(dt2 / (2 * frames) * 1e3, 3),
                                      "note": "host input and host output as above, two frames in flight (2 contexts / streams / host threads)"}
            out = outs[0]
        except Exception as e:
            line["pcie_pipelined"] = {"value": None, "note": "failed: %r" % (e,)}
        # BASELINE.json configs[1] (1 M uniform-random particles) measured in the same run, HBM-resident like `value`
        if args.workload != "s1m":
            try:
                w1 = W.WORKLOADS["s1m"]
                p1 = Parameters(particle_radius=w1["particle_radius"], compact_support_radius=np.float32(2.0 * w1["smoothing_length"] * w1["particle_radius"]),
                                cube_size=np.float32(w1["cube_size"] * w1["particle_radius"]), auto_disable=False)
                d1 = torch.from_numpy(w1["gen"]()).to(dev)
                o1 = None
                for _ in range(2):
                    o1 = ctx.reconstruct(d1, p1, out=o1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    o1 = ctx.reconstruct(d1, p1, out=o1)
                torch.cuda.synchronize()
                dt1 = (time.perf_counter() - t1) / args.steps
                line["other_configs"] = {"s1m": {"value": round(d1.shape[0] / dt1 / 1e6, 3), "unit": "Mparticles/s", "ms_per_step": round(dt1 * 1e3, 3),
                                                 "n_particles": int(d1.shape[0]), "n_vertices": int(o1.stats["n_vertices"]),
                                                 "note": "BASELINE.json configs[1]: 1 M uniform-random particles in the unit cube, r=0.01, cell=1.0"}}
                del d1, o1
            except Exception as e:
                line["other_configs"] = {"s1m": {"value": None, "note": "failed: %r" % (e,)}}
    if not sharded_path:
        # HBM traffic of the splat kernel from rocprofv3 PMC passes (collected offline, see profiles/)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "splat_traffic.json"))).get(args.workload)
            if tr and tr.get("kernel", "").startswith(line["roofline"]["kernel"]):
                line["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                line["roofline"]["traffic_note"] = tr["note"]
                if tr.get("valu_insts_per_launch") and line["roofline"]["kernel_ms"] > 0:
                    # the kernel's real bound (informative): share of the VALU issue slots it uses, 1024 SIMDs, one wave64
                    # instruction per 2 cycles, at the device's maximum engine clock
                    props = torch.cuda.get_device_properties(dev)
                    clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
                    n_simd = 4 * int(props.multi_processor_count)
                    slots = line["roofline"]["kernel_ms"] * 1e-3 * clock_hz * n_simd / 2.0
                    line["roofline"]["valu"] = {"insts_per_launch": tr["valu_insts_per_launch"], "issue_slots_frac": round(tr["valu_insts_per_launch"] / slots, 4),
                                                "clock_ghz": round(clock_hz * 1e-9, 3), "simds": n_simd, "note": tr.get("valu_note", "")}
        except Exception:
            pass
    if sharded_path and getattr(last, "timings", None):
        line["sharded_step_ms"] = {k: round(v / args.steps, 3) for k, v in last.timings.items()}
    if rank == 0:
        if not sharded_path and not args.no_cpu_baseline and not args.main_only:
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_sample_scale)
            except Exception as e:  # the baseline is informative; never lose the measurement because of it
                line["cpu_baseline"] = {"value": None, "unit": "Mparticles/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
