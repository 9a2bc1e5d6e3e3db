#!/bin/bash
# tools/build_asan.sh: the debug build of the library with AddressSanitizer into splashsurf_amd/variants/libsplashsurf_hip_asan.so
# (git-ignored; travels to the GPU box).  Default: the HOST code of the library is instrumented (-fsanitize=address -fno-gpu-sanitize:
# buffer bookkeeping, exchange packing, accessors, the C ABI), the kernels are the release kernels at -O1.  DEVICE=1 also instruments
# the kernels (gfx950:xnack+); that build needs the sanitizer flavour of the HIP runtime (/opt/rocm/lib/asan) on the box: without it
# the device reports cannot be delivered ("Hostcall: no handler found for service ID 4") and device allocations carry no shadow.
# This image has no /opt/rocm/lib/asan, so the device flavour builds here but cannot run.
# Run a workload under the build with tools/asan_check.sh, which preloads the sanitizer runtime and runs the workload twice,
# comparing every output bit for bit with each other and with the release build (the race check of SURVEY.md section 5).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/splashsurf_amd/csrc"
OBJ=$(mktemp -d)
ARCH=gfx950; GPUSAN=-fno-gpu-sanitize
if [ "$DEVICE" == "1" ]; then ARCH=gfx950:xnack+; GPUSAN=; fi
FLAGS="--offload-arch=$ARCH -fsanitize=address $GPUSAN -shared-libsan -g -O1 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-pass-failed"
pids=()
for f in ss_api ss_kernels ss_global ss_post ss_dist ss_prims ss_pipeline; do
  hipcc $FLAGS -c $f.hip -o "$OBJ/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
mkdir -p "$ROOT/splashsurf_amd/variants"
hipcc --offload-arch=$ARCH -fsanitize=address $GPUSAN -shared-libsan -shared -fPIC "$OBJ"/ss_api.o "$OBJ"/ss_kernels.o "$OBJ"/ss_global.o "$OBJ"/ss_post.o "$OBJ"/ss_dist.o "$OBJ"/ss_prims.o "$OBJ"/ss_pipeline.o \
  -ldl -lpthread -o "$ROOT/splashsurf_amd/variants/libsplashsurf_hip_asan.so"
rm -rf "$OBJ"
echo "built splashsurf_amd/variants/libsplashsurf_hip_asan.so"
