#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "attention_core or generations or block_against or baseline_head or full_size_block or full_size_attention" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log; grep -E "^E  " $O/tests.log | head -5
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s p50", round(d["step_ms"]["p50"],3), o.get("attn_bwd"), o.get("attn_fwd"))'
B="python bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
SWV2_ATTN_BWD8=0 $B 2>/dev/null | python -c "$pick" nopos_bwd11
$B 2>/dev/null | python -c "$pick" nopos_bwd8
SWV2_ATTN_BWD8=0 $B --rel-pos 1 2>/dev/null | python -c "$pick" relpos_bwd11
$B --rel-pos 1 2>/dev/null | python -c "$pick" relpos_bwd8
done
