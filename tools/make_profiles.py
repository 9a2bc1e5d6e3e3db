#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/r02prof/) into the tracked artefacts under profiles/:
  r02_s10m_tank_simd{0,1,2}_kernel_stats.csv   `rocprofv3 --kernel-trace --stats` of `bench.py --main-only --steps 10 --warmup 2 --simd M`
  r02_pmc_s10m_tank.md                          PMC counters per launch of the splat / density kernels (separate passes)
  splat_traffic.json                            HBM bytes and VALU instructions per launch of the splat kernel (k_splat_fused), read by bench.py
FETCH_SIZE / WRITE_SIZE are reported in KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's FETCH_SIZE tallies
the 128-B requests of wide (16 B per lane) streaming reads at 64 B, so the read side is doubled; WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TAG = sys.argv[2] if len(sys.argv) > 2 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else TAG + "prof")
DST = os.path.join(ROOT, "profiles")
MODES = {0: "scalar", 1: "simd", 2: "simd_hw"}


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n


def main():
    traffic = {"s10m_tank": {}}
    md = ["# rocprofv3 PMC counters, S10M-tank, 1x MI355X (%s)" % TAG,
          "Command per pass: `rocprofv3 --pmc <COUNTERS> --kernel-include-regex 'k_splat|k_density_sub|k_mc_|k_rs_|k_chained_scan' --output-format csv -- python bench.py --main-only --steps 1 "
          "--warmup 1 --simd M` (tools/collect_profiles.sh; one pass per counter group, no --kernel-trace/--stats in a PMC pass).  Values are per launch "
          "(two launches per run agree to 4 digits).  FETCH_SIZE / WRITE_SIZE in KiB as reported; `hbm` = 2 x FETCH_SIZE + WRITE_SIZE in bytes "
          "(gfx950 tallies the 128-B requests of 16-B-per-lane streaming reads at 64 B, MI355X_MICROARCH.md).", ""]
    for m, mname in MODES.items():
        stats = os.path.join(SRC, "stats_simd%d" % m, "run_kernel_stats.csv")
        shutil.copyfile(stats, os.path.join(DST, TAG + "_s10m_tank_simd%d_kernel_stats.csv" % m))
        dur = {}
        for r in csv.DictReader(open(stats)):
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6)
        vals = collections.defaultdict(dict)
        for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            for r in csv.DictReader(open(os.path.join(SRC, "pmc_simd%d_%s" % (m, tag), "run_counter_collection.csv"))):
                vals[short(r["Kernel_Name"])].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        md.append("## enable_simd = %d (%s)" % (m, mname))
        md.append("| kernel | avg ms (kernel stats, %d calls) | FETCH_SIZE KiB | WRITE_SIZE KiB | hbm bytes | hbm rate | SQ_INSTS_VALU | SQ_INSTS_SALU | SQ_INSTS_LDS | SQ_WAVES | VALU insts per SIMD-cycle pair |" % 12)
        md.append("|---|---|---|---|---|---|---|---|---|---|---|")
        for k in sorted(vals):
            v = {c: sum(x) / len(x) for c, x in vals[k].items()}
            hbm = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024 + v.get("WRITE_SIZE", 0.0) * 1024
            ms = dur.get(k, (0, 0.0))[1]
            rate = hbm / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            slots = ms * 1e-3 * 2.4e9 * 1024 / 2.0 if ms > 0 else 0.0
            md.append("| %s | %.3f | %.6g | %.6g | %.4g | %.2f TB/s | %.4g | %.4g | %.4g | %.4g | %.3f |" % (
                k, ms, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), hbm, rate, v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0),
                v.get("SQ_WAVES", 0), v.get("SQ_INSTS_VALU", 0) / slots if slots else 0.0))
            if k.startswith("k_splat_accumulate") or k.startswith("k_splat_fused"):
                t = traffic["s10m_tank"].setdefault(mname, {
                    "kernel": "k_splat_fused (first and second pass) + k_splat_accumulate_list (blocks with over 192 candidates)", "hbm_bytes_per_launch": 0.0, "fetch_size_bytes_reported": 0.0,
                    "write_size_bytes": 0.0, "kernel_ms_rocprof_avg": 0.0, "valu_insts_per_launch": 0.0, "launches": {},
                    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-include-regex 'k_splat|k_density_sub|k_mc_|k_rs_|k_chained_scan') on S10M-tank, per step = sum over the "
                            "launches of the splat kernel; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 128-B requests as 64 B); the "
                            "kernel gathers every block's candidates from the cell-sorted particle array (neighbouring blocks re-read the same rows, mostly "
                            "from L2) and writes the level-set values of the evaluated sub-blocks (DESIGN.md section 5, profiles/" + TAG + "_pmc_s10m_tank.md)",
                    "valu_note": "SQ_INSTS_VALU of both launches per step (profiles/" + TAG + "_pmc_s10m_tank.md); a SIMD-32 issues one wave64 VALU instruction per 2 cycles at best"})
                t["hbm_bytes_per_launch"] += hbm
                t["fetch_size_bytes_reported"] += v.get("FETCH_SIZE", 0.0) * 1024
                t["write_size_bytes"] += v.get("WRITE_SIZE", 0.0) * 1024
                t["kernel_ms_rocprof_avg"] += ms
                t["valu_insts_per_launch"] += v.get("SQ_INSTS_VALU", 0.0)
                t["launches"][k] = {"ms_rocprof_avg": ms, "hbm_bytes": hbm, "valu_insts": v.get("SQ_INSTS_VALU", 0.0)}
                g = {c: sum(x) / len(x) for c, x in vals.get("k_splat_gather<float>", {}).items()}
                t["other_kernels"] = {"k_splat_gather<float>": {"fetch_size_bytes_reported": g.get("FETCH_SIZE", 0.0) * 1024, "write_size_bytes": g.get("WRITE_SIZE", 0.0) * 1024,
                                                                "hbm_bytes_per_launch": 2.0 * g.get("FETCH_SIZE", 0.0) * 1024 + g.get("WRITE_SIZE", 0.0) * 1024}}
        md.append("")
    open(os.path.join(DST, TAG + "_pmc_s10m_tank.md"), "w").write("\n".join(md) + "\n")
    stamp_file = os.path.join(SRC, "kernel_source_stamp.txt")  # written by tools/collect_profiles.sh from the sources that were profiled
    traffic["_collected_from"] = {"kernel_source_stamp": open(stamp_file).read().strip() if os.path.exists(stamp_file) else None, "round": TAG,
                                  "note": "bench.py attaches these numbers only to a build with the same stamp (sha256 of ss_kernels.hip, ss_device.h, ss_api.hip, ss_prims.h, ss_prims.hip)"}
    json.dump(traffic, open(os.path.join(DST, "splat_traffic.json"), "w"), indent=1)
    print("\n".join(md[-40:]))


if __name__ == "__main__":
    main()
