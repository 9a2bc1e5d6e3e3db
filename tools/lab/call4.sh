#!/bin/bash
# r05 call 4: packed-f16 certificate walk as the only walk of the fused kernel (68 VGPRs, 7 waves per SIMD like the f32 build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05d; mkdir -p $O
V=$PWD/splashsurf_amd/variants
for rep in 1 2; do
  for lib in "" pk1 pk2 pk1r62 pk1r58; do
    SPLASHSURF_HIP_LIB=${lib:+$V/libsplashsurf_hip_$lib.so} timeout 300 python tools/ab_kernels.py --workload s10m_tank --simd 0 --steps 8 --digest --tag "${lib:-f32walk}" >> $O/ab.jsonl 2>> $O/ab.err
  done
done
echo "== prof1" >> $O/phase.txt
SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_prof1.so timeout 300 python tools/phase_prof.py --simd 0 >> $O/phase.txt 2>> $O/phase.err
SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_pk1.so timeout 300 python -m pytest tests/test_gpu_certificates.py -x -q 2>&1 | tail -3 >> $O/cert_pk1.log
python - <<'PY'
import json
for l in open('gpurun_out/r05d/ab.jsonl'):
    d=json.loads(l); print(d['tag'], round(d['ms_total'],3), round(d['ms_levelset_accumulate'],3), d['certified_frac'], d['digest'])
PY
tail -12 $O/phase.txt; cat $O/cert_pk1.log
