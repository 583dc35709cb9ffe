#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "rollout or multistep or ddp_two_ranks or ddp_bucket" 2>&1 | tail -3
for i in 1 2; do
python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
SWV2_ROLLOUT_DSKIP_FUSE=0 python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
done
