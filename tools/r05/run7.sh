#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
python tools/trainer_rate.py bench_depth12_e128 100 2>/dev/null | grep Trainer | tee $O/trainer_rate.txt
python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('bench.py loop', round(d['value'],1), 'samples/s, p50 step', round(d['step_ms']['p50'],3))" | tee -a $O/trainer_rate.txt
bash tools/trace_cfg.sh r05_cfg4 bench_geo_depth24_e192_invar 2 6 > $O/trace_cfg4.txt 2>&1; head -40 $O/trace_cfg4.txt
