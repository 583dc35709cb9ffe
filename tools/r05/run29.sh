#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_round.sh r05 2>&1 | tail -45
bash tools/pmc_wait.sh r05 2>&1 | tail -16
bash tools/trace_bench.sh r05_relpos --rel-pos 1 --no-secondary 2>&1 | tail -25
bash tools/pmc_relpos.sh r05 2>&1 | tail -12
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; tail -c 600 gpurun_out/r05_bench.json
