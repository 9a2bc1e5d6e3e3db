"""The library's HIP kernels, run WITHOUT a GPU: the `-m gpu` parity tests themselves in a child process whose SPLASHSURF_HIP_LIB names
tests/emu/_build/libsplashsurf_emu.so -- splashsurf_amd/csrc/*.hip compiled for the host against tests/emu's CPU execution model of the HIP
kernel language (wave64 fibers, ballots, DPP, MFMA, permlane swap, chained look-back; tests/emu/include/hip/hip_runtime.h).  Same sources,
same C ABI, same tests, same oracle and wheel goldens: what this tier adds to the CPU suite is the kernels' LOGIC (indexing, wave-level
protocols, summation order, certificates), which so far only ran on the GPU box.  What it cannot show: anything about time, occupancy or
the memory system, and the two approximate device instructions (v_sqrt_f32 in the certificates' records and in enable_simd = 2).

Test infrastructure only: the emulated library is never loaded by the product (api.py needs SPLASHSURF_HIP_LIB set explicitly, as here).
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

# -m gpu tests that cannot run on the emulated library.  (The C++ host of test_cpp_host_over_c_abi links against whatever SPLASHSURF_HIP_LIB names.)  (Tests that hand the library torch tensors hand it HOST tensors here: conftest.device_name().)
NEED_A_DEVICE = []  # (none left: SPLASHSURF_EMU_HOST_DEVICE makes tests/conftest.py hand out the host where host code asks torch for a "cuda" device)
# ... and the ones that take more than ~4 s emulated (8 host threads); SPLASHSURF_EMU_ALL=1 runs them too (about 20 minutes, 1 M particles
# included; the 10 M / 40 M full-size tests stay out)
SLOW_EMULATED = [
    "tests/test_gpu_parity.py::test_hbm_bandwidth_probe_reports_plausible_rates",
    "tests/test_gpu_parity.py::test_full_size_s10m_tank_bit_identical_to_oracle",
    "tests/test_gpu_parity.py::test_config4_s40m_tank",
    "tests/test_gpu_dist_native.py::test_native_full_s40m_tank_four_ranks",
]
SLOW_EMULATED_OPTIONAL = [
    "tests/test_gpu_prims.py::test_radix_sort_is_a_stable_sort[20000001-24-1]",
    "tests/test_gpu_prims.py::test_radix_sort_is_a_stable_sort[3000001-25-1]",
    "tests/test_gpu_prims.py::test_radix_sort_is_a_stable_sort[3000001-32-0]",
    "tests/test_gpu_prims.py::test_radix_sort_is_a_stable_sort[1000003-24-0]",
    "tests/test_gpu_prims.py::test_radix_sort_is_a_stable_sort[1000003-23-1]",
    "tests/test_gpu_prims.py::test_chained_scan_equals_cumsum[20000001]",
    "tests/test_gpu_parity.py::test_gpu_global_strategy_bit_identical_to_oracle_and_reference[global_free_particles_125]",
    "tests/test_gpu_parity.py::test_u64_triangles_cross_pcie_as_u32",
    "tests/test_gpu_simd.py::test_simd_bit_identical_to_uniform_oracle[simd_config2_s1m]",
    "tests/test_gpu_simd.py::test_simd_bit_identical_to_uniform_oracle[simd_config5_hilbert]",
    "tests/test_gpu_fuzz.py::test_random_configuration_bit_identical[clusters-global-c0.33-n7-f64]",
    "tests/test_gpu_fuzz.py::test_random_configuration_bit_identical[clusters-grid-c0.2-n64-f32]",
    "tests/test_gpu_fuzz.py::test_random_configuration_bit_identical[clusters-global-c0.33-n9-f32]",
    "tests/test_gpu_parity.py::test_gpu_matches_reference_digest[config2_s1m]",
    "tests/test_gpu_parity.py::test_gpu_matches_reference_digest[config5_hilbert]",
    "tests/test_gpu_parity.py::test_gpu_bit_identical_to_oracle_large[config5_hilbert]",
    "tests/test_gpu_parity.py::test_dense_cloud_exceeding_tile_capacity[True]",
    "tests/test_gpu_parity.py::test_split_mc_offsets_gives_the_same_mesh",
    "tests/test_gpu_simd.py::test_early_exit_inside_the_fluid_changes_no_output[0]",
    "tests/test_gpu_dist_native.py::test_native_brick_resident_time_series_ships_halos_only",
    "tests/test_gpu_parity.py::test_host_waits_are_counted",
    "tests/test_gpu_simd.py::test_early_exit_inside_the_fluid_changes_no_output[1]",
    "tests/test_gpu_simd.py::test_early_exit_inside_the_fluid_changes_no_output[2]",
    "tests/test_gpu_simd.py::test_simd_matches_reference_simd_digest[simd_config2_s1m]",
    "tests/test_gpu_simd.py::test_simd_matches_reference_simd_digest[simd_config5_hilbert]",
    "tests/test_gpu_simd.py::test_simd_hw_sqrt_within_the_references_own_tolerance[simd_config5_hilbert]",
    "tests/test_gpu_dist_native.py::test_native_partition_feedback_keeps_the_mesh",
    "tests/test_gpu_dist_native.py::test_native_ranks_reproduce_single_context[tank_crop-8-float32-1]",
    "tests/test_gpu_dist_native.py::test_native_ranks_reproduce_single_context[hilbert_n32-8-float32-1]",
    "tests/test_gpu_dist_native.py::test_native_ranks_reproduce_single_context[hilbert_n32-4-float32-0]",
    "tests/test_gpu_certificates.py::test_certified_subblocks_lie_inside_the_fluid[tank_bulk_scalar]",
    "tests/test_gpu_certificates.py::test_certified_subblocks_lie_inside_the_fluid[tank_bulk_simd]",
    "tests/test_gpu_certificates.py::test_certified_subblocks_lie_inside_the_fluid[tank_fine_grid]",
    "tests/test_gpu_certificates.py::test_certified_subblocks_lie_inside_the_fluid[tank_large_units]",
    "tests/test_reference_suite.py::test_full_rs[free_particles_02]",
    "tests/test_reference_suite.py::test_full_rs[free_particles_01]",
    "tests/test_reference_suite.py::test_subdomains_rs_single_particle[0.025-tris2-verts2-subdomains2]",
]
FILES = ["tests/test_gpu_prims.py", "tests/test_gpu_parity.py", "tests/test_gpu_simd.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_certificates.py", "tests/test_gpu_dist_native.py",
         "tests/test_reference_suite.py", "tests/test_post.py", "tests/test_distributed.py", "tests/test_cli.py", "tests/test_gpu_pipeline.py"]


def emulated_library():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return build_emu.build()


def run_gpu_tests_emulated(extra_args, deselect, timeout_s):
    lib = emulated_library()
    import build_emu
    env = dict(os.environ, SPLASHSURF_HIP_LIB=lib, SPLASH_RCCL_LIB=build_emu.build_fake_rccl())  # (the one-rank RCCL test binds the stand-in)
    env["SPLASHSURF_EMU_HOST_DEVICE"] = "1"
    env.setdefault("HIP_EMU_THREADS", "2")  # four pytest workers with two emulator threads each: most cases are small and bound by launch latency, not by cores
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "600", "-n", "4"] + FILES + list(extra_args)
    for d in deselect:
        cmd += ["--deselect", d]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
    tail = "\n".join(p.stdout.splitlines()[-40:])
    m = re.search(r"(\d+) passed", p.stdout)
    return p.returncode, (int(m.group(1)) if m else 0), tail


def test_the_gpu_parity_tests_pass_on_the_cpu_execution_model_of_the_kernels():
    everything = os.environ.get("SPLASHSURF_EMU_ALL") == "1"
    deselect = NEED_A_DEVICE + SLOW_EMULATED + ([] if everything else SLOW_EMULATED_OPTIONAL)
    rc, passed, tail = run_gpu_tests_emulated([], deselect, 7200 if everything else 1500)
    assert rc == 0, tail
    assert passed >= 205, tail  # the scan / sort primitives, 50 parity / golden cases, 52 fuzz cases, the SIMD modes, certificates, in-process ranks, the reference's own test cases


def test_the_rccl_branch_between_rank_processes_with_a_stand_in_rccl():
    """tests/test_gpu_dist_rccl.py for world 2 and 4 on the smallest case: one PROCESS per rank (torch.distributed.run), each with the emulated
    library, ss_comm_create_rccl bound to tests/emu/fake_rccl.cpp through SPLASH_RCCL_LIB -- the branch of ss_dist.hip a one-GPU box never
    takes (count matrices, offsets, grouped ncclSend / ncclRecv per peer, the small collectives), checked against the single-context mesh bit for
    bit.  The stand-in verifies that every receive finds its message with the announced size and type; what real RCCL does on real links stays
    unmeasured.  (All twelve cases -- three clouds, world 1 / 2 / 4 / 8 -- pass this way; the 1.2 M-particle ones take minutes each.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    lib, fake = build_emu.build(), build_emu.build_fake_rccl()
    env = dict(os.environ, SPLASHSURF_HIP_LIB=lib, SPLASH_RCCL_LIB=fake, SPLASH_EMULATED_RANKS="1", HIP_EMU_THREADS="2")
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "600", "tests/test_gpu_dist_rccl.py", "-k",
           "test_rccl_ranks and dam_break and (f64-0-2 or f64-0-4)"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    tail = "\n".join(p.stdout.splitlines()[-40:])
    assert p.returncode == 0 and re.search(r"\b2 passed", p.stdout), tail


def test_outputs_do_not_depend_on_the_schedule():
    """Race check: the same five reconstructions (tools/emu_schedule_digest.py: fine and coarse grid, both arithmetics, certification forced, an
    over-dense cube) under the default schedule and under HIP_EMU_SHUFFLE -- workgroups in a scrambled order, the waves of a workgroup and the
    lanes between two synchronisation points in reverse order, all schedules the device may produce -- must give one digest."""
    lib = emulated_library()
    digests = []
    for extra in ({}, {"HIP_EMU_SHUFFLE": "1"}, {"HIP_EMU_SHUFFLE": "2", "HIP_EMU_THREADS": "3"}):
        env = dict(os.environ, SPLASHSURF_HIP_LIB=lib, **extra)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu_schedule_digest.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:]
        digests.append(p.stdout.split()[-2])
    assert len(set(digests)) == 1 and len(digests[0]) == 24, digests


def test_launch_trace_of_a_small_job():
    """What one steady-state call on BASELINE config 1 dispatches (tools/emu_launch_trace.py over HIP_EMU_TRACE): a small job's time is its list of
    dispatches and host waits, and the list is the same on the GPU (same host flow).  Budgets = the values of the build the round-6 numbers were taken
    from; every memset clears whole 16-byte units (ss_round16: an unaligned size costs ROCm a second fill launch -- 10 fill launches for 6 memsets in
    profiles/r06_cfg_pmc.md)."""
    lib = emulated_library()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import emu_launch_trace
    old = os.environ.get("SPLASHSURF_HIP_LIB")
    os.environ["SPLASHSURF_HIP_LIB"] = lib
    try:
        t = emu_launch_trace.trace("config1")
        t5 = emu_launch_trace.trace("config5")
    finally:
        if old is None:
            del os.environ["SPLASHSURF_HIP_LIB"]
        else:
            os.environ["SPLASHSURF_HIP_LIB"] = old
    assert (t["n_vertices"], t["n_triangles"]) == (33026, 66220)
    assert t["launches"] <= 37, t["kernels"]
    assert len(t["memsets"]) <= 6 and all(m % 16 == 0 for m in t["memsets"]), t["memsets"]
    assert t["copies"] == [4732 * 12]  # the upload; no other copy (counts reach the host through mail slots)
    assert t["n_host_waits"] <= 8
    # BASELINE config 5 (no over-dense blocks: the arena path is not launched)
    assert (t5["n_vertices"], t5["n_triangles"]) == (533960, 1067920)
    assert t5["launches"] <= 31 and len(t5["memsets"]) <= 6 and all(m % 16 == 0 for m in t5["memsets"]) and t5["n_host_waits"] <= 7, t5


def test_graft_entry_smoke_on_the_cpu_execution_model():
    """__graft_entry__.smoke() -- the driver's first call on the GPU box -- with the emulated library in the HIP build's place: config 1 bit-identical to the oracle."""
    env = dict(os.environ, SPLASHSURF_HIP_LIB=emulated_library())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "[smoke] ok: 33026 vertices / 66220 triangles" in p.stdout, p.stdout[-2000:]


def _bench_dry_run(args, timeout_s=1200):
    import json
    lib = emulated_library()
    env = dict(os.environ, SPLASHSURF_HIP_LIB=lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "bench_dry_run.py")] + list(args), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout_s)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line
    return json.loads(lines[0])


def test_bench_py_default_record_dry_run():
    """bench.py's own code, unchanged, on the CPU execution model (tests/emu/bench_dry_run.py: torch's CUDA entry points pointed at the host, workloads of a few
    thousand particles): the default N = 1 record is complete -- contract keys, roofline, every extra (arithmetic modes, host-to-host frames, the frame
    pipeline, the post-processing recipe, the other configurations, the coarse-grid splat) -- and no extra reports a failure.  The driver's bench run is the
    one artifact nobody can re-run when the GPU pool is closed; this keeps a typo in it from costing the round's measurement."""
    d = _bench_dry_run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "Mparticles/s end-to-end reconstruct" and d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f32" and d["dry_run"] is True
    assert d["config"]["workload"] == "s10m_tank" and d["config"]["splat_two_pass"] == 1 and d["enable_simd"] == 0
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert k in d["roofline"], k
    failed = []

    def walk(x, path):
        if isinstance(x, dict):
            if ("value" in x and x["value"] is None and path != "") or str(x.get("note", "")).startswith("failed"):
                failed.append((path, str(x)[:300]))
            for k, v in x.items():
                walk(v, path + "/" + k)

    for k in ("arithmetic_modes", "e2e_host_u64", "pcie_inclusive", "pcie_pipelined", "pcie_pipelined_u64", "post_pipeline", "other_configs", "splat_hbm_bound"):
        assert k in d, k
        walk(d[k], k)
    assert not failed, failed
    assert set(d["other_configs"]) == {"s1m", "s10m_cube", "s40m_tank_1gpu", "config1_dam_break", "config5_hilbert"}
    assert d["pcie_pipelined"]["n_vertices_per_frame"] == d["config"]["n_vertices"] == d["pcie_pipelined_u64"]["n_vertices_per_frame"]
    assert d["other_configs"]["config1_dam_break"]["n_vertices"] == 33026  # (the real input file: the wheel golden's mesh size)


def test_bench_py_pseudo_rank_record_dry_run():
    """... and the sharded code path of bench.py (--pseudo-ranks: ss_dist_* over an in-process group), with the contiguous slices and with brick-resident
    particles (the default; --slices for the former): same mesh either way, a complete projection object."""
    a = _bench_dry_run(["--pseudo-ranks", "2", "--steps", "1", "--warmup", "1", "--main-only", "--slices"])
    b = _bench_dry_run(["--pseudo-ranks", "2", "--steps", "1", "--warmup", "1", "--main-only"])
    assert a["config"]["resident"] is False and b["config"]["resident"] is True
    for d in (a, b):
        assert d["pseudo_ranks"] == 2 and len(d["per_rank"]) == 2 and d["scaling"] == "strong"
        for k in ("own_ms_slowest_rank", "projected_step_ms", "projected_step_ms_point_to_point_links", "collective_steps"):
            assert k in d["projection"], k
        assert sum(r["owned_particles"] for r in d["per_rank"]) == d["config"]["n_particles"]
    assert (a["config"]["n_vertices_incl_shared"], a["config"]["n_triangles"]) == (b["config"]["n_vertices_incl_shared"], b["config"]["n_triangles"])
    assert b["exchange"]["bytes_sent_per_step_all_ranks"] <= a["exchange"]["bytes_sent_per_step_all_ranks"]


def _bench_ranks_dry_run(world, extra, port):
    """The driver's own command line for N > 1 (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...), with tests/emu/bench_dry_run.py in bench.py's place: one PROCESS per rank, each with the emulated library (HIP_EMU_DEVICES = N: every rank
    selects its LOCAL_RANK), RCCL = the stand-in of tests/emu/fake_rccl.cpp, torch.distributed over gloo."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    env = dict(os.environ, SPLASHSURF_HIP_LIB=build_emu.build(), SPLASH_RCCL_LIB=build_emu.build_fake_rccl(), HIP_EMU_THREADS="2", HIP_EMU_DEVICES=str(world))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "emu", "bench_dry_run.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1"] + list(extra)
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{\"metric\"")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_py_multi_gpu_command_line_dry_run():
    """bench.py --gpus N as the driver launches it, N = 2 and 4, never run on more than one real GPU: rank processes, the library's RCCL exchange path
    (stand-in RCCL), the strong-scaling reference on rank 0, the per-rank table; with brick-resident particles (the default), with contiguous slices
    (--slices) and over the torch.distributed fallback (--exchange torch) as well.  All variants give the same mesh."""
    base = _bench_ranks_dry_run(2, [], 29621)
    assert base["n_gpus"] == 2 and base["scaling"] == "strong" and base["config"]["workload"] == "s40m_tank" and len(base["per_rank"]) == 2
    assert base["exchange"]["kind"].startswith("native") and base["exchange"]["rccl_world_size"] == 2
    assert base["single_gpu_same_workload"]["value"] and base["single_gpu_same_workload"]["speedup_of_this_run"] > 0
    mesh = (base["config"]["n_vertices"], base["config"]["n_triangles"])
    assert base["resident"] is True  # (the default: every rank holds its brick's particles)
    res = _bench_ranks_dry_run(2, ["--main-only", "--slices"], 29622)
    assert res["resident"] is False and (res["config"]["n_vertices"], res["config"]["n_triangles"]) == mesh
    tor = _bench_ranks_dry_run(2, ["--main-only", "--exchange", "torch"], 29623)
    assert tor["exchange"]["kind"].startswith("torch.distributed") and (tor["config"]["n_vertices"], tor["config"]["n_triangles"]) == mesh
    four = _bench_ranks_dry_run(4, ["--main-only"], 29624)
    assert four["n_gpus"] == 4 and len(four["per_rank"]) == 4 and four["exchange"]["rccl_world_size"] == 4
    assert four["config"]["n_triangles"] == mesh[1]  # (triangles are disjoint between ranks; shared face vertices are counted by every holder)
    assert sum(r["owned_particles"] for r in four["per_rank"]) == four["config"]["n_particles"]


def test_the_emulated_library_is_not_what_the_product_loads():
    """api.library_path() names the HIP build unless SPLASHSURF_HIP_LIB says otherwise; nothing under splashsurf_amd/, bench.py or
    __graft_entry__.py mentions the emulator."""
    import splashsurf_amd.api as A
    if "SPLASHSURF_HIP_LIB" not in os.environ:
        assert A.library_path().endswith(os.path.join("splashsurf_amd", "libsplashsurf_hip.so"))
    hits = []
    for base, _, names in os.walk(os.path.join(ROOT, "splashsurf_amd")):
        if "csrc" + os.sep + "build" in base or "__pycache__" in base or "variants" in base:
            continue
        for n in names:
            if n.endswith((".py", ".hip", ".h", ".hpp")):
                text = open(os.path.join(base, n), errors="replace").read()
                if "libsplashsurf_emu" in text or "tests/emu/_build" in text:
                    hits.append(os.path.join(base, n))
    for n in ("bench.py", "__graft_entry__.py"):
        if "libsplashsurf_emu" in open(os.path.join(ROOT, n)).read():
            hits.append(n)
    assert not hits, hits
