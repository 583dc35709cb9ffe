#!/bin/bash
cd $GRAFT_REPO_ROOT
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", round(d["step_ms"]["p50"],3), "wgrad_group(+reduce)", o.get("wgrad_group"))'
B="python bench.py --no-cpu-baseline --no-secondary"
for M in "" "-DSWV2_SLAB_ABL=4" "-DSWV2_SLAB_ABL=8" "-DSWV2_SLAB_ABL=12"; do
  SO=/tmp/libswv2_ab.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $M -o $SO swin_v2_weather_amd/csrc/*.hip 2>/dev/null
  SWV2_LIB=$SO $B 2>/dev/null | python -c "$pick" "variant[$M]"
done
