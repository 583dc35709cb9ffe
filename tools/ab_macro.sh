#!/bin/bash
# Same-box A/B of a compile-time variant: builds a private copy of the library with the given -D macros, then alternates
# bench.py runs with the shipped and the private library (box-to-box variance is +-1.5 %, a same-box pair is needed for
# small effects).  Usage on the GPU box: tools/ab_macro.sh "-DSWV2_TN_GELU_ABL" [bench args...]
set -e
MACROS="$1"; shift
SO=/tmp/libswv2_ab.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $MACROS -o $SO swin_v2_weather_amd/csrc/*.hip 2>/dev/null
pick='import json,sys
d=json.loads(sys.stdin.readline())
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s", o)'
for i in 1 2; do
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" shipped
  SWV2_LIB=$SO python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$pick" "variant[$MACROS]"
done
