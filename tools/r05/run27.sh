#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
pick='import json,sys
d=json.loads(sys.stdin.readline())
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", round(d["step_ms"]["p50"],3), o)'
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "attn or attention or block or fixture" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "$pick" new
  SWV2_LIB=$R/tools/r05/_so/libswv2_old.so python bench.py --no-cpu-baseline --no-secondary $1 2>/dev/null | python -c "$pick" old
done
