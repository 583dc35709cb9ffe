#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cpb or stage_level or fused_mlp or cfg4 or baseline_head" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/trace_bench.sh r05_relpos3 --rel-pos 1 --no-secondary > $O/trace_relpos.txt 2>&1
grep -E "cpb|pack|distribution|attn_" gpurun_out/r05_relpos3_kernel_stats.md
python tools/run_cfg.py bench_geo_depth24_e192_invar 2 8 2>/dev/null | tail -1
python tools/run_cfg.py bench_geo_depth24_e192_invar 2 8 2>/dev/null | tail -1
bash tools/batch_fit.sh > $O/batch_fit.txt 2>&1; tail -40 $O/batch_fit.txt
