#!/bin/bash
# Developer tool: build a variant of libsvr_hip.so into build_ab/ with extra compiler flags, for
# tools/ab_libs.py (interleaved A/B of several builds in one process on one box).
#   tools/build_variant.sh NAME [-DSOMETHING ...]   ->  build_ab/libsvr_hip_NAME.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/simple-vk-renderer_amd/csrc
out=$root/build_ab/obj_$name
mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter $*"
pids=()
for f in svr_api k_geometry k_flatten k_bin k_tile k_image; do
  /opt/rocm/bin/hipcc $flags -c "$src/$f.hip" -o "$out/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build_ab/libsvr_hip_$name.so" "$out"/*.o
rm -rf "$out"
echo "built build_ab/libsvr_hip_$name.so"
