#!/bin/bash
# r05 call 2: packed-f16 certificate walk (in-tree) against the f32 walk (variant f32walk): soundness test, full GPU suite, A/B timings with digests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_certificates.py -x -q 2>&1 | tail -15 > $O/pytest_cert.log
tail -3 $O/pytest_cert.log
for simd in 0 1; do
  AB_ARGS="--simd $simd --steps 8" bash tools/ab_kernels.sh $O/ab_simd$simd.jsonl s10m_tank s10m_tank > /dev/null 2>&1
done
AB_ARGS="--simd 0 --steps 5" bash tools/ab_kernels.sh $O/ab_other.jsonl s1m s10m_cube config5 > /dev/null 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_all.log
tail -3 $O/pytest_all.log
