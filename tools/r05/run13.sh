#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "attention_core" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
SECONDS=0
python bench.py > gpurun_out/r05_bench.json 2>$O/bench.err; echo "bench wall ${SECONDS}s"
python -c "
import json; d=json.load(open('gpurun_out/r05_bench.json')); print('bench', round(d['value'],1), d['step_ms']['p50'], d['roofline']['frac'], d['roofline']['traffic'], d['attention_module']['mfma_pipe_busy_frac'], [ (s.get('value') and round(s['value'],1)) for s in d['secondary']])"
