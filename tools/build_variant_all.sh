#!/bin/bash
# Build container: a private variant of the WHOLE library with extra macros (all sources recompiled in parallel into /tmp).
# usage: tools/build_variant_all.sh <tag> "<-D macros>"   -> swin_v2_weather_amd/libswv2_<tag>.so (select with SWV2_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
TAG=$1; MACROS=$2
D=/tmp/variant_all_$TAG; mkdir -p $D
SRCS=$(python3 -c "from swin_v2_weather_amd import _lib as L; print(' '.join(L.SOURCES))")
for s in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $MACROS -c swin_v2_weather_amd/csrc/$s -o $D/$s.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o swin_v2_weather_amd/libswv2_$TAG.so $D/*.o
echo swin_v2_weather_amd/libswv2_$TAG.so
