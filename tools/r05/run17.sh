#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "rollout or multistep or ddp_bucket_view or ddp_two_ranks" 2>&1 | tail -3
python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
SWV2_GRAD_ACC_INPLACE=0 python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
SWV2_GRAD_ACC_INPLACE=0 python tools/run_cfg.py bench_depth12_e128_2step 2 8 2>/dev/null | tail -1
