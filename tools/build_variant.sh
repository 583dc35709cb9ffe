#!/bin/bash
# Build container: a private variant of the library with extra macros for ONE source file, linked against the other objects of the
# last build (swin_v2_weather_amd/build/*.o).  The .so travels to the GPU box with gpurun; select it with SWV2_LIB=<path>.
# usage: tools/build_variant.sh <source.hip> <tag> "<-D macros>"
set -e
cd "$(dirname "$0")/.."
SRC=$1; TAG=$2; MACROS=$3
O=/tmp/variant_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $MACROS -c swin_v2_weather_amd/csrc/$SRC -o $O
OBJS=$(python3 -c "from swin_v2_weather_amd import _lib as L; print(' '.join('swin_v2_weather_amd/build/' + s + '.o' for s in L.SOURCES if s != '$SRC'))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o swin_v2_weather_amd/libswv2_$TAG.so $OBJS $O
echo swin_v2_weather_amd/libswv2_$TAG.so
