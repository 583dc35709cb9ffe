#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
pick='import json,sys
d=json.loads(sys.stdin.readline())
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", d["step_ms"]["p50"], o)'
for v in p1 p4; do
  SWV2_LIB=$R/tools/r05/_so/libswv2_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "attn or attention" 2>&1 | tail -1
done
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$pick" shipped
  for v in p1 p4; do
    SWV2_LIB=$R/tools/r05/_so/libswv2_$v.so python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "$pick" $v
  done
done
