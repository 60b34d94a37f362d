#!/bin/bash
# Development: build a variant of the library with extra -D flags for some of its source files and link it beside the
# real one:
#   tools/build_variant.sh NAME wino "-DAVID_W_LATE=-1"          ->  avid-cma_amd/avid_hip/libavid_hip_NAME.so
#   tools/build_variant.sh w2fp32 "conv wino" "-DAVID_W2_FP32"   (a flag that more than one file must see)
# and run with AVID_HIP_LIB=/root/repo/avid-cma_amd/avid_hip/libavid_hip_NAME.so (A/B on one GPU box).
set -e
cd "$(dirname "$0")/../avid-cma_amd"
make -j8 > /dev/null
name=$1; srcs=$2; flags=$3
mkdir -p build/var
for src in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c csrc/$src.hip -o build/var/${src}_$name.o &
done
wait
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  if [[ " $srcs " == *" $b "* ]]; then objs="$objs build/var/${b}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o avid_hip/libavid_hip_$name.so $objs
echo built avid_hip/libavid_hip_$name.so
