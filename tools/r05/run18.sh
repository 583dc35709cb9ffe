#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "attn or attention or relpos or rel_pos or bias or cpb or fixture" 2>&1 | tail -4
bash tools/ab_macro.sh "-DSWV2_BIAS_BWD_ROLLED" --rel-pos 1 --no-secondary 2>&1 | tail -4
