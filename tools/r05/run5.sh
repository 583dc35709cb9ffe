#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cpb or stage_level or device_pool or ddp_two_ranks or trainer_end" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log; grep "DDP buckets" $O/tests.log
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
o={k["kernel"]:round(k["avg_ms"]*1e3,1) for k in [d["roofline"]]+d.get("roofline_others",[]) if "kernel" in k}
print(sys.argv[1], round(d["value"],1), "samples/s", d["step_ms"]["p50"], d["step_ms"]["sequence"][:2], o)'
B="python bench.py --no-cpu-baseline --no-secondary"
$B --rel-pos 1 2>/dev/null | python -c "$pick" relpos
SWV2_ATTN_FWD3B_KREG=1 $B --rel-pos 1 2>/dev/null | python -c "$pick" relpos_kreg
$B --rel-pos 1 2>/dev/null | python -c "$pick" relpos
SWV2_ATTN_FWD3B_KREG=1 $B --rel-pos 1 2>/dev/null | python -c "$pick" relpos_kreg
bash tools/trace_bench.sh r05_relpos2 --rel-pos 1 --no-secondary > $O/trace_relpos.txt 2>&1
cp gpurun_out/r05_relpos2_kernel_stats.md $O/
bash tools/batch_fit.sh > $O/batch_fit.txt 2>&1; tail -45 $O/batch_fit.txt
