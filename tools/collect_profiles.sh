#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace statistics and PMC passes of the profiled bench command, per arithmetic
# mode (Parameters::enable_simd = 0 / 1 / 2).  Counters are collected in their own passes, without --kernel-trace/--stats
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).  Output: gpurun_out/${ROUND:-r05}prof/.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/${ROUND:-r05}prof
rm -rf $OUT; mkdir -p $OUT
python -c "import bench; print(bench.kernel_source_stamp())" > $OUT/kernel_source_stamp.txt
for m in ${MODES:-0 1 2}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_simd$m -o run -- python bench.py --main-only --steps 10 --warmup 2 --simd $m > $OUT/bench_simd$m.log 2>&1
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" ${EXTRA_PMC:+"$EXTRA_PMC"}; do
    tag=$(echo $c | cut -d' ' -f1)
    rocprofv3 --pmc $c --kernel-include-regex 'k_splat|k_density_sub|k_mc_|k_rs_|k_chained_scan' --output-format csv -d $OUT/pmc_simd${m}_$tag -o run -- python bench.py --main-only --steps 1 --warmup 1 --simd $m > $OUT/pmc_simd${m}_$tag.log 2>&1
  done
done
find $OUT -name "*.csv" | head -40
