#!/bin/bash
# r05 call 5: ablation timings of the radix sort pass (which part of k_rs_pass does a pass wait for?)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05e; mkdir -p $O
V=$PWD/splashsurf_amd/variants
for lib in "" abl1 abl2 abl3 abl4 abl5; do
  t=${lib:-base}
  SPLASHSURF_HIP_LIB=${lib:+$V/libsplashsurf_hip_$lib.so} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$t -o run -- python tools/lab/sort_only.py > $O/$t.log 2>&1
  f=$(find $O/$t -name "run_kernel_stats.csv" | head -1)
  echo "== $t"; grep -E "k_rs_pass|k_rs_hist" $f | cut -d, -f1-4 | cut -c1-40,170-
done
