#!/usr/bin/env python3
"""tools/make_cfg_profiles.py RAW_DIR TAG -- turns the raw rocprofv3 output of tools/profile_configs.sh (gpurun_out/<TAG>prof_cfg/<config>/...) into the tracked
artefacts under profiles/: <TAG>_cfg_<config>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `tools/ab_kernels.py` on that configuration) and
<TAG>_cfg_pmc.md (per kernel and configuration: average duration, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md, VALU / SALU / LDS instructions, the share
of the VALU issue slots, waves, SQ_WAIT_ANY per wave-cycle)."""
import collections
import csv
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1])
TAG = sys.argv[2]
DST = os.path.join(ROOT, "profiles")
DESC = {"s10m_cube": "S10M-cube (BASELINE config 3 read literally: 10 M uniform-random particles in the unit cube, 10x over-dense, R = 8)",
        "s1m": "S1M (BASELINE config 2: 1 M uniform-random particles, r = 0.01, cube size 1.0 r, R = 4)",
        "r2": "R = 2: the S10M-tank particles at cube size 2 r (33 grid points per particle: the HBM-bound splat configuration of the bench line)",
        "config1": "config 1 (double dam break, 4 732 particles)", "config5": "config 5 (hilbert, 46 843 particles, cube size 0.45 r)"}


def short(n):
    return n.split("(")[0].replace("void ", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rp::")


def main():
    md = ["# rocprofv3 kernel statistics and PMC counters of the secondary configurations, 1x MI355X (%s)" % TAG,
          "Commands: `tools/profile_configs.sh` -- per configuration one `rocprofv3 --kernel-trace --stats` pass and three `--pmc` passes (FETCH_SIZE; WRITE_SIZE; the SQ group) of "
          "`python tools/ab_kernels.py --workload ...`, counters in their own passes without --kernel-trace/--stats.  Durations are the kernel-stats averages; counters are per launch.  "
          "`hbm` = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; gfx950 tallies the 128-B requests of 16-B-per-lane reads at 64 B).  `issue` = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x "
          "2.4 GHz x duration).", ""]
    for cfg in sorted(os.listdir(SRC)):
        stats = os.path.join(SRC, cfg, "stats", "run_kernel_stats.csv")
        if not os.path.exists(stats):
            continue
        shutil.copyfile(stats, os.path.join(DST, "%s_cfg_%s_kernel_stats.csv" % (TAG, cfg)))
        dur = {}
        rows = list(csv.DictReader(open(stats)))
        for r in rows:
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) * 1e-6, float(r["Percentage"]))
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        for t in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            f = os.path.join(SRC, cfg, "pmc_%s" % t, "run_counter_collection.csv")
            if os.path.exists(f):
                for r in csv.DictReader(open(f)):
                    vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        md.append("## %s" % DESC.get(cfg, cfg))
        md.append("| kernel | calls | avg ms | share | hbm bytes | hbm rate | SQ_INSTS_VALU | issue | SQ_INSTS_SALU | SQ_INSTS_LDS | SQ_WAVES | SQ_WAIT_ANY / SQ_WAVE_CYCLES |")
        md.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
        for k, (calls, ms, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:22]:
            v = {a: sum(b) / len(b) for a, b in vals.get(k, {}).items()}
            hbm = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024
            slots = ms * 1e-3 * 2.4e9 * 1024 / 2.0
            md.append("| %s | %d | %.4f | %.1f %% | %s | %s | %s | %s | %s | %s | %s | %s |" % (
                k[:90], calls, ms, pct, ("%.4g" % hbm) if v else "", ("%.2f TB/s" % (hbm / (ms * 1e-3) / 1e12)) if v and ms > 0 else "",
                ("%.4g" % v["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in v else "", ("%.2f" % (v["SQ_INSTS_VALU"] / slots)) if "SQ_INSTS_VALU" in v and slots else "",
                ("%.4g" % v["SQ_INSTS_SALU"]) if "SQ_INSTS_SALU" in v else "", ("%.4g" % v["SQ_INSTS_LDS"]) if "SQ_INSTS_LDS" in v else "",
                ("%.4g" % v["SQ_WAVES"]) if "SQ_WAVES" in v else "", ("%.2f" % (v["SQ_WAIT_ANY"] / max(v.get("SQ_WAVE_CYCLES", 1.0), 1.0))) if "SQ_WAIT_ANY" in v else ""))
        log = os.path.join(SRC, cfg, "stats.log")
        if os.path.exists(log):
            last = [l for l in open(log) if l.startswith("{")]
            if last:
                md.append("")
                md.append("stage timers of the profiled run (`tools/ab_kernels.py`): `%s`" % last[-1].strip())
        md.append("")
    open(os.path.join(DST, TAG + "_cfg_pmc.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md[:60]))


if __name__ == "__main__":
    main()
