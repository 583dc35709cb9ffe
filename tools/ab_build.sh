#!/bin/bash
# GPU box: build private variants of the library with -D macros and run a probe with each.
# usage: tools/ab_build.sh "<probe command>" "<macros A>" "<macros B>" ...     (an empty string = the shipped flags)
PROBE="$1"; shift
i=0
for M in "$@"; do
  SO=/tmp/libswv2_ab$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $M -o $SO swin_v2_weather_amd/csrc/*.hip 2>/dev/null
  echo "== variant [$M]"
  SWV2_LIB=$SO $PROBE 2>&1 | grep -v amdgpu.ids
  i=$((i+1))
done
