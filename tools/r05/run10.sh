#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ddp_two_ranks" > $O/tests.log 2>&1; echo "ddp test rc=$?"; grep "DDP buckets" $O/tests.log
bash tools/profile_round.sh r05 > $O/profile_round.txt 2>&1; tail -20 $O/profile_round.txt
bash tools/trace_bench.sh r05_relpos --rel-pos 1 --no-secondary > $O/trace_relpos.txt 2>&1
bash tools/pmc_wait.sh r05 > $O/pmc_wait.txt 2>&1; tail -16 $O/pmc_wait.txt
python bench.py > gpurun_out/r05_bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench.json')); print('bench', round(d['value'],1), d['step_ms']['p50'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], [ (s.get('value') and round(s['value'],1)) for s in d['secondary']])"
python bench.py --no-cpu-baseline --no-secondary --rel-pos 1 > gpurun_out/r05_bench_relpos.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --local-batch 8 > gpurun_out/r05_bench_b8.json 2>/dev/null
python -c "
import json
for f in ('relpos','b8'):
    d=json.load(open('gpurun_out/r05_bench_%s.json'%f)); print(f, round(d['value'],1), d['step_ms']['p50'])"
