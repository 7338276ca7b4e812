#!/bin/bash
# Alternative build of the library for A/B runs (tools/gpu_ab.py): tools/build_ab.sh <name> "<extra hipcc flags>" [waves]
# -> ab_libs/librtmi_<name>.so (render_kernel.hip recompiled with the flags, the other objects taken from the tree)
set -e
NAME=$1; EXTRA=$2; WAVES=${3:-7}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/ray-tracing-in-cuda_amd/csrc
mkdir -p $R/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result \
  -DRT_WAVES_PER_SIMD=$WAVES -DRT_GROUP=4 -DRTMI_ABLATIONS=1 $EXTRA -c -o $R/ab_libs/rk_$NAME.o $S/render_kernel.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab_libs/librtmi_$NAME.so $S/scene.o $S/capi.o $S/render_host.o $R/ab_libs/rk_$NAME.o $S/tiles.o
echo built ab_libs/librtmi_$NAME.so
