#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "loss or head or model or trajectory or curve" 2>&1 | tail -3
python tools/probe_head.py 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), d['step_ms']['p50'])"
