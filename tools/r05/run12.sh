#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
/usr/bin/time -v python bench.py > gpurun_out/r05_bench.json 2>$O/bench.err; grep -E "Elapsed|Maximum resident" $O/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench.json')); print('bench', round(d['value'],1), d['step_ms']['p50'], d['roofline']['frac'], d['roofline']['traffic'], d['attention_module']['mfma_pipe_busy_frac'], [ (s.get('value') and round(s['value'],1)) for s in d['secondary']])"
