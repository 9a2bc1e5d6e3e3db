#!/bin/bash
# tools/build_variant.sh NAME [GIT_REF] [-- extra hipcc flags]: builds splashsurf_amd/variants/libsplashsurf_hip_NAME.so from the csrc/ of
# GIT_REF (default: the working tree) with extra flags (e.g. -DSS_TUNE_X=1), reusing the objects of the files that are not kernels.
# Variants are loaded with SPLASHSURF_HIP_LIB=... (splashsurf_amd/api.py) to time kernel changes side by side on one GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
REF=""
if [ -n "$1" ] && [ "$1" != "--" ]; then REF=$1; shift; fi
[ "$1" == "--" ] && shift
W=$(mktemp -d)
if [ -n "$REF" ]; then
  git -C "$ROOT" archive "$REF" splashsurf_amd/csrc include | tar -x -C "$W"
else
  mkdir -p "$W/splashsurf_amd" && cp -r "$ROOT/splashsurf_amd/csrc" "$W/splashsurf_amd/" && cp -r "$ROOT/include" "$W/"
fi
cd "$W/splashsurf_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function"
pids=()
for f in ss_api ss_kernels ss_global ss_post ss_dist ss_prims ss_pipeline; do
  [ -e $f.hip ] || continue
  hipcc $FLAGS "$@" -c $f.hip -o $f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
mkdir -p "$ROOT/splashsurf_amd/variants"
hipcc --offload-arch=gfx950 -shared -fPIC $(ls ss_api.o ss_kernels.o ss_global.o ss_post.o ss_dist.o ss_prims.o ss_pipeline.o 2>/dev/null) -ldl -lpthread -o "$ROOT/splashsurf_amd/variants/libsplashsurf_hip_$NAME.so"
rm -rf "$W"
echo "built splashsurf_amd/variants/libsplashsurf_hip_$NAME.so"
