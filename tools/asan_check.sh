#!/bin/bash
# tools/asan_check.sh OUT_DIR [workload]: runs one workload (default tank_small, both kernel families) twice through the AddressSanitizer
# build of the library (tools/build_asan.sh) on the GPU box and compares the digests of densities / vertices / triangles of the two runs
# and of the release build: the debug target of SURVEY.md section 5 (sanitizer report = memory error; differing digests = race).
#   gpurun -- 'bash tools/asan_check.sh gpurun_out/asan'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
OUT=${1:-gpurun_out/asan}; WL=${2:-tank_small}
mkdir -p "$OUT"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
: > "$OUT/asan_check.jsonl"
for simd in 0 1; do
  timeout 120 python tools/ab_kernels.py --workload $WL --steps 1 --warmup 1 --simd $simd --digest --host --tag release >> "$OUT/asan_check.jsonl" 2>> "$OUT/asan_check.err"
  for run in 1 2; do
    LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 \
      SPLASHSURF_HIP_LIB=$ROOT/splashsurf_amd/variants/libsplashsurf_hip_asan.so \
      timeout 300 python tools/ab_kernels.py --workload $WL --steps 1 --warmup 1 --simd $simd --digest --host --tag asan_run$run >> "$OUT/asan_check.jsonl" 2>> "$OUT/asan_check.err"
    echo "simd=$simd run=$run exit=$?" >> "$OUT/asan_check.err"
  done
done
python - "$OUT/asan_check.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
ok = True
for simd in (0, 1):
    d = {r["tag"]: r.get("digest") for r in rows if r["simd"] == simd}
    same = len(d) == 3 and len(set(d.values())) == 1
    ok &= same
    print("simd=%d digests %s -> %s" % (simd, d, "identical" if same else "DIFFER / missing"))
print("asan_check:", "ok" if ok else "FAILED")
PY
tail -30 "$OUT/asan_check.err"
