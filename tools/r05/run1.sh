#!/bin/bash
# round 5, GPU call 1: the tests the CPB hoist / bias forward touch, then same-box A/Bs of the rel_pos=True step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "cpb or attention_core or stage_level or block_against or baseline_head or whole_model or loss_curve_relpos or droppath or generations" > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
B="python bench.py --no-cpu-baseline --no-secondary"
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s", d["step_ms"], o)'
$B 2>$O/b0.err | tee $O/bench_nopos.json | python -c "$pick" nopos
$B --rel-pos 1 2>$O/b1.err | tee $O/bench_relpos.json | python -c "$pick" relpos_hoisted_fwd3b
SWV2_ATTN_FWD3B=0 $B --rel-pos 1 2>$O/b2.err | tee $O/bench_relpos_nofwd3b.json | python -c "$pick" relpos_hoisted_firstgen_fwd
SWV2_CPB_PER_BLOCK=1 $B --rel-pos 1 2>$O/b3.err | tee $O/bench_relpos_perblock.json | python -c "$pick" relpos_perblock_fwd3b
SWV2_CPB_PER_BLOCK=1 SWV2_ATTN_FWD3B=0 $B --rel-pos 1 2>$O/b4.err | tee $O/bench_relpos_r04path.json | python -c "$pick" relpos_r04_path
tail -3 $O/b1.err
