#!/bin/bash
# Build a variant of the library in the build container: recompile ONE source with extra -D macros, link with the shipped objects.
# usage: tools/build_variant.sh attn.hip "-DSWV2_BWD_PIPE=2" tools/r05/_so/libswv2_p2.so   (select it with SWV2_LIB=...)
set -e
cd "$(dirname "$0")/.."
SRC=$1; MACROS=$2; OUT=$3
mkdir -p "$(dirname "$OUT")" /tmp/swv2_variant
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $MACROS -c swin_v2_weather_amd/csrc/$SRC -o /tmp/swv2_variant/$SRC.o
OBJS=$(ls swin_v2_weather_amd/build/*.hip.o | grep -v "/$SRC.o" | grep -v attn_bwd8)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS /tmp/swv2_variant/$SRC.o
echo "built $OUT"
