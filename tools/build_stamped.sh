#!/bin/bash
# Build container: a stamped variant of the library (attention backward traces) with extra macros.  usage: tools/build_stamped.sh "<macros>" out.so
set -e
cd "$(dirname "$0")/.."
mkdir -p "$(dirname "$2")" /tmp/swv2_stamped
for f in attn.hip attn_bwd_stream.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSWV2_ATTNS_STAMPS -DSWV2_ATTN1_STAMPS $1 -c swin_v2_weather_amd/csrc/$f -o /tmp/swv2_stamped/$f.o &
done
wait
OBJS=$(ls swin_v2_weather_amd/build/*.hip.o | grep -v "/attn.hip.o" | grep -v "/attn_bwd_stream.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$2" $OBJS /tmp/swv2_stamped/attn.hip.o /tmp/swv2_stamped/attn_bwd_stream.hip.o
echo "built $2"
