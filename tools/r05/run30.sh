#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "loss or head or rollout or multistep or model or curve" 2>&1 | tail -2
python tools/probe_head.py 2>&1 | tail -4
