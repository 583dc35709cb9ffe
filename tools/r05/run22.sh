#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "loss or head or curve or trajectory" 2>&1 | tail -2
python tools/probe_head.py 2>&1 | grep "loss epi"
for v in noy h3; do SWV2_LIB=$R/tools/r05/_so/libswv2_$v.so python tools/probe_head.py 2>&1 | grep "loss epi"; done
