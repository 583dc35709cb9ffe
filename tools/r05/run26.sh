#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "rollout or multistep or loss or head" 2>&1 | tail -15
for i in 1 2; do
python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
SWV2_LOSS_IN_HEAD_ROLLOUT=0 python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
done
