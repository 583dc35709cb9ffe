#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "cfg4 or mlp" 2>&1 | tail -2
for i in 1 2; do
python tools/run_cfg.py bench_geo_depth24_e192_invar 2 6 2>/dev/null | tail -1
SWV2_MLP_FWD192=2 python tools/run_cfg.py bench_geo_depth24_e192_invar 2 6 2>/dev/null | tail -1
done
