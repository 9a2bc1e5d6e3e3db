#!/bin/bash
# r05 call 3: certificate walk in packed f16 -- SoA slots (pk1), AoS + v_perm (pk2), the same in one asm statement (pk3) against the f32 walk (in-tree);
# phase profiles of the three formulations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c; mkdir -p $O
V=$PWD/splashsurf_amd/variants
for rep in 1 2; do
  for lib in "" pk1 pk2 pk3; do
    SPLASHSURF_HIP_LIB=${lib:+$V/libsplashsurf_hip_$lib.so} timeout 300 python tools/ab_kernels.py --workload s10m_tank --simd 0 --steps 8 --digest --tag "${lib:-f32walk}" >> $O/ab.jsonl 2>> $O/ab.err
  done
done
for lib in prof0 prof1 prof2; do
  echo "== $lib" >> $O/phase.txt
  SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_$lib.so timeout 300 python tools/phase_prof.py --simd 0 >> $O/phase.txt 2>> $O/phase.err
done
SPLASHSURF_HIP_LIB=$V/libsplashsurf_hip_pk3.so timeout 300 python -m pytest tests/test_gpu_certificates.py -x -q 2>&1 | tail -3 >> $O/cert_pk3.log
cat $O/phase.txt | tail -40
