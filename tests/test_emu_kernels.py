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

# -m gpu tests that cannot run on the emulated library: host code of theirs asks torch for a CUDA device.  (The C++ host of
# test_cpp_host_over_c_abi links against whatever SPLASHSURF_HIP_LIB names.)  (Tests that hand the library torch tensors hand it HOST tensors here: conftest.device_name().)
NEED_A_DEVICE = [
    "tests/test_cli.py::test_cli_end_to_end_matches_the_library_call",
    "tests/test_post.py::test_gpu_pipeline_matches_oracle_and_reference",
]
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
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "--timeout", "600"] + FILES + list(extra_args)
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
    assert passed >= 183, tail  # the scan / sort primitives, 50 parity / golden cases, 52 fuzz cases, the SIMD modes, certificates, in-process ranks, the reference's own test cases


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
