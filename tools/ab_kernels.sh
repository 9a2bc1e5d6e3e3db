#!/bin/bash
# tools/ab_kernels.sh OUT WORKLOADS... : every library variant under splashsurf_amd/variants/ and the in-tree build through tools/ab_kernels.py
# (run on the GPU box; OUT gets one JSON line per (variant, workload)).  AB_ARGS: extra arguments for ab_kernels.py.
# Name a workload twice for a control: runs of one build differ by about +-0.05 ms on the S10M-tank splat kernel, and the first process on a
# fresh box has measured up to 2 % slow.
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
for wl in "$@"; do
  for lib in "" splashsurf_amd/variants/*.so; do
    if [ -n "$lib" ] && [ ! -e "$lib" ]; then continue; fi
    SPLASHSURF_HIP_LIB=${lib:+$PWD/$lib} timeout 300 python tools/ab_kernels.py --workload "$wl" --digest $AB_ARGS >> "$OUT" 2>> "$OUT.err" || echo "{\"tag\": \"$lib\", \"workload\": \"$wl\", \"failed\": true}" >> "$OUT"
  done
done
cat "$OUT"
