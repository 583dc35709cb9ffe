#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
SECONDS=0
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench wall $SECONDS s"
python bench.py --no-cpu-baseline --no-secondary --rel-pos 1 > gpurun_out/r05_bench_relpos.json 2>/dev/null
python bench.py --no-cpu-baseline --no-secondary --local-batch 8 > gpurun_out/r05_bench_b8.json 2>/dev/null
python - <<'PY'
import json
for f in ("r05_bench","r05_bench_relpos","r05_bench_b8"):
    d=json.loads(open(f"gpurun_out/{f}.json").readline())
    print(f, round(d["value"],1), d["ms_per_step"], d["step_ms"]["p50"], d["roofline"]["frac"], d["roofline"].get("traffic"), [ (s["workload"][:24], round(s["value"],1)) for s in d.get("secondary",[])])
PY
bash tools/batch_fit.sh 2>&1 | tail -34 > gpurun_out/r05_batch_fit.txt; tail -3 gpurun_out/r05_batch_fit.txt
python tools/trainer_rate.py 2>&1 | tail -6 | tee gpurun_out/r05_trainer_rate.txt
