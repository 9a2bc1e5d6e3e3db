#!/bin/bash
# r05 call 7: AddressSanitizer build of the library (host code instrumented): reconstruct twice + the rewritten multi-GPU host flow and post-processing under it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g; mkdir -p $O
bash tools/asan_check.sh $O/tank_small tank_small > $O/asan_tank_small.txt 2>&1
bash tools/asan_check.sh $O/s1m s1m > $O/asan_s1m.txt 2>&1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 SPLASHSURF_HIP_LIB=$PWD/splashsurf_amd/variants/libsplashsurf_hip_asan.so \
  timeout 900 python -m pytest tests/test_gpu_dist_native.py tests/test_post.py -m gpu -x -q -k "not s40m" > $O/asan_pytest_dist_post.txt 2>&1
echo "pytest exit=$?" >> $O/asan_pytest_dist_post.txt
grep -c "AddressSanitizer" $O/*.txt $O/*/asan_check.err
grep -E "asan_check:|passed|failed|exit=" $O/*.txt | tail -12
