#!/bin/bash
cd $GRAFT_REPO_ROOT
SWV2_ATTN_FWD3_R2=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "attention_core or generations" 2>&1 | tail -2
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", round(d["step_ms"]["p50"],3), "attn_fwd", o.get("attn_fwd"))'
B="python bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
$B 2>/dev/null | python -c "$pick" fwd3
SWV2_ATTN_FWD3_R2=1 $B 2>/dev/null | python -c "$pick" fwd3_r2
done
