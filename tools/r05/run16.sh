#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block_against" 2>&1 | tail -2
bash tools/trace_cfg.sh r05_cfg5 bench_depth12_e128_2step 2 6 > $O/trace_cfg5.txt 2>&1; head -44 $O/trace_cfg5.txt
