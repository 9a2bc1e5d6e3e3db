#!/bin/bash
# tests/emu/run_asan.sh [LOG] [pytest args ...] -- AddressSanitizer over the KERNELS: the -m gpu tests on the emulated library built with -fsanitize=address
# (tests/emu/build_emu.py --asan; buffers at their exact requested sizes, ss_host.h).  The GPU pool has no device sanitizer; this is the
# nearest thing: every load / store of every kernel checked against the hipMalloc'ed extents, on the CPU.  About 15x slower than the plain emulation.
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT" || exit 1
LOG=${1:-/tmp/emu_asan.log}; shift
LIB=$(python tests/emu/build_emu.py --asan | tail -1)
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then
  ARGS=(tests/test_gpu_parity.py tests/test_gpu_simd.py tests/test_gpu_fuzz.py tests/test_gpu_certificates.py tests/test_gpu_dist_native.py tests/test_reference_suite.py
        -k "not full_size and not s40m and not s1m and not hbm and not rccl and not cpp_host and not device_pointer and not sharded_engine and not grid_loop_fixture and not host_waits and not config5 and not free_particles_125 and not u64_triangles and not tank_bulk and not tank_crop")
fi
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:detect_stack_use_after_return=0:abort_on_error=0 SPLASHSURF_HIP_LIB=$LIB \
  python -m pytest -m gpu -q -p no:cacheprovider --timeout 1800 "${ARGS[@]}" > "$LOG" 2>&1
echo "exit=$?" >> "$LOG"
grep -c "ERROR: AddressSanitizer" "$LOG" | sed 's/^/AddressSanitizer reports: /' >> "$LOG"
tail -5 "$LOG"
