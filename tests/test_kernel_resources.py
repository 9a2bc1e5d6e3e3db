"""Register / scratch / LDS budgets of the hot kernels (CPU: hipcc's kernel-resource-usage remarks of the gfx950 build, no GPU needed).

The measured numbers of DESIGN.md rest on an occupancy: k_splat_fused at 7 waves per SIMD (<= 72 VGPRs, 5 104 bytes of LDS per one-wave workgroup, no
scratch), the density kernel and the arena kernels at 8, four workgroups of k_splat_certify_big per CU (<= 40 960 bytes of LDS).  A change that
pushes a kernel over one of these lines changes its speed without any test noticing -- least of all in a container without a GPU; this test notices.
The bounds are the values of the build the round-6 profiles were taken from (tools/kernel_resources.sh prints the table)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

# substring of the mangled name -> (max VGPRs, max scratch bytes per lane, max LDS bytes per workgroup)
BUDGETS = {
    "k_splat_fusedIfLi0ELb1E": (72, 0, 5120), "k_splat_fusedIfLi1ELb1E": (72, 0, 5120), "k_splat_fusedIfLi3ELb1E": (72, 0, 5120), "k_splat_fusedIfLi4ELb1E": (72, 0, 5120),
    "k_splat_fusedIfLi0ELb0E": (64, 0, 5120), "k_splat_fusedIfLi1ELb0E": (64, 0, 5120), "k_splat_fusedIfLi3ELb0E": (64, 0, 5120), "k_splat_fusedIfLi4ELb0E": (64, 0, 5120),
    "k_splat_fusedIdLi0ELb1E": (104, 0, 7168), "k_splat_fusedIdLi0ELb0E": (96, 0, 7168),
    "k_splat_certify_big": (64, 64, 40960),
    "k_density_subIfLi0ELb1E": (64, 0, 16384), "k_density_subIfLi0ELb0E": (64, 0, 16384), "k_density_subIfLi1ELb1E": (64, 0, 16384), "k_density_subIdLi0ELb0E": (64, 0, 16384),
    "k_splat_accumulate_listIfLi1ELb0E": (64, 0, 40960), "k_splat_accumulate_listIfLi1ELb1E": (64, 64, 40960),
    "k_splat_gatherIfE": (64, 0, 16384), "k_splat_gather_largeIfLi8192E": (64, 0, 40960),
    "k_mc_countIfE": (32, 0, 4096), "k_mc_emitIfE": (48, 0, 20480), "k_mc_neighboursIfE": (32, 0, 0),
    "k_mark_blocksIfE": (32, 0, 0), "k_emit_copiesIfE": (40, 0, 0), "k_cell_keysIfE": (32, 0, 0), "k_make_posvolIfE": (16, 0, 0),
    "k_chained_scanIj8SSOpPlusLi8192ELi512E12SSClassifyInIfE": (72, 0, 1024), "k_chained_scanIj8SSOpPlusLi8192ELi512E10SSMcFlagInIfE": (72, 0, 1024),
    "k_chained_scanIj7SSOpMaxLi8192ELi512E13SSCellTableIn": (64, 0, 1024),
}


@pytest.fixture(scope="module")
def resources():
    import __graft_entry__ as G
    return G.kernel_resources("ss_kernels.hip")


def test_hot_kernels_stay_within_their_register_scratch_and_lds_budgets(resources):
    assert len(resources) > 100  # every instantiation of the translation unit reports
    missing, over = [], []
    for key, (vgprs, scratch, lds) in BUDGETS.items():
        hits = [(n, r) for n, r in resources.items() if key in n]
        if not hits:
            missing.append(key)
            continue
        for n, r in hits:
            if r["vgprs"] > vgprs or r["scratch"] > scratch or r["lds"] > lds:
                over.append((n[:70], r, (vgprs, scratch, lds)))
    assert not missing, "kernels the budget table names do not exist any more: %s" % missing
    assert not over, "over budget (VGPRs, scratch bytes per lane, LDS bytes): %s" % over


def test_no_kernel_of_the_f32_path_spills_unless_listed(resources):
    """Scratch is HBM traffic the roofline does not count: only the two kernels pinned at 64 registers for 8 waves per SIMD may spill (measured that way)."""
    allowed = ("k_splat_certify_big", "k_splat_accumulate_list")
    spills = {n[:70]: r["scratch"] for n, r in resources.items() if r["scratch"] > 0 and not any(a in n for a in allowed)}
    assert not spills, spills


def test_the_sort_pass_and_the_other_translation_units():
    """k_rs_pass: 112 VGPRs (4 waves per SIMD) and 43 044 bytes of LDS per 512-thread tile (three tiles per CU) is the measured configuration; nothing in
    ss_prims / ss_post / ss_global / ss_dist spills."""
    import __graft_entry__ as G
    prims = G.kernel_resources("ss_prims.hip")
    for n, r in prims.items():
        if "k_rs_passILi512E" in n:
            assert r["vgprs"] <= 128 and r["scratch"] == 0 and r["lds"] <= 49152, (n, r)
    for tu in ("ss_prims.hip", "ss_post.hip", "ss_global.hip", "ss_dist.hip"):
        spills = {n[:70]: r["scratch"] for n, r in G.kernel_resources(tu).items() if r["scratch"] > 0}
        assert not spills, (tu, spills)
