#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
pick='import json,sys
d=json.loads(sys.stdin.readline())
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", round(d["step_ms"]["p50"],3), o)'
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "attn or attention or relpos or rel_pos or bias or cpb or fixture" 2>&1 | tail -2
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-secondary --rel-pos 1 2>/dev/null | python -c "$pick" stream
  SWV2_ATTN_FWD3B_STREAM=0 python bench.py --no-cpu-baseline --no-secondary --rel-pos 1 2>/dev/null | python -c "$pick" regs
done
