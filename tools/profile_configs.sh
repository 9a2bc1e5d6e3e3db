#!/bin/bash
# tools/profile_configs.sh TAG [CONFIG...]: runs on the GPU box (gpurun).  rocprofv3 kernel-trace statistics and PMC passes of the secondary
# configurations -- the ones whose level set goes through the over-dense ("arena") path of the splat, and the small jobs:
#   s10m_cube  BASELINE config 3 read literally (10x over-dense)      s1m  BASELINE config 2
#   r2         the S10M-tank particles at cube size 2 r (R = 2, the HBM-bound splat configuration)
#   config1 / config5  the two data-file configs (fixed cost per call)
# Counters are collected in their own passes without --kernel-trace/--stats.  Output: gpurun_out/${TAG}prof_cfg/<config>/...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r04}; shift
CONFIGS=${@:-s10m_cube s1m r2}
OUT=gpurun_out/${TAG}prof_cfg
mkdir -p $OUT
for c in $CONFIGS; do
  case $c in
    s10m_cube) ARGS="--workload s10m_cube --steps 4 --warmup 2";;
    s1m) ARGS="--workload s1m --steps 10 --warmup 2";;
    r2) ARGS="--workload s10m_tank --cube-size 2.0 --steps 6 --warmup 2";;
    config1) ARGS="--workload config1 --steps 20 --warmup 3";;
    config5) ARGS="--workload config5 --steps 20 --warmup 3";;
    *) ARGS="--workload $c --steps 6 --warmup 2";;
  esac
  rm -rf $OUT/$c; mkdir -p $OUT/$c
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c/stats -o run -- python tools/ab_kernels.py $ARGS > $OUT/$c/stats.log 2>&1
  if [ -z "$NO_PMC" ]; then
    for p in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
      t=$(echo $p | cut -d' ' -f1)
      timeout 600 rocprofv3 --pmc $p --kernel-include-regex 'k_splat|k_density_sub|k_mc_|k_rs_|k_chained_scan' --output-format csv -d $OUT/$c/pmc_$t -o run -- python tools/ab_kernels.py $ARGS --steps 1 --warmup 1 > $OUT/$c/pmc_$t.log 2>&1
    done
  fi
  tail -1 $OUT/$c/stats.log
done
find $OUT -name "*kernel_stats.csv" | head
