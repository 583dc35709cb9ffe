#!/bin/bash
# GPU box: the round's measurement set on the final sources, one call -> gpurun_out/ (copy into profiles/ afterwards: profiles/README.md).  usage: tools/final_measure.sh r06
T=${1:-r06}
tools/profile_round.sh $T > gpurun_out/prof_$T.txt 2>&1
tools/pmc_wait.sh $T > gpurun_out/pmcw_$T.txt 2>&1
cp gpurun_out/prof_$T/${T}_pmc_hbm.json gpurun_out/prof_$T/${T}_pmc_mfma.json profiles/       # bench.py quotes counter traffic from profiles/ when the source hash matches
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tools/round_aux.sh $T 2>&1 | tail -8
tools/trace_cfg.sh ${T}_cfg4 bench_geo_depth24_e192_invar 2 6 > /dev/null 2>&1
tools/pmc_cfg.sh ${T}_cfg4 bench_geo_depth24_e192_invar 2 4 > /dev/null 2>&1
tools/trace_cfg.sh ${T}_cfg5 bench_depth12_e128_2step 2 6 > /dev/null 2>&1
tools/pmc_cfg.sh ${T}_cfg5 bench_depth12_e128_2step 2 4 > /dev/null 2>&1
tools/trace_bench.sh ${T}_relpos --rel-pos 1 --no-secondary > /dev/null 2>&1
tools/pmc_relpos.sh $T > /dev/null 2>&1
tools/batch_fit.sh > gpurun_out/${T}_batch_fit.txt 2>&1; tail -1 gpurun_out/${T}_batch_fit.txt
python tools/trainer_rate.py > gpurun_out/${T}_trainer_rate.txt 2>&1; tail -2 gpurun_out/${T}_trainer_rate.txt
tail -2 gpurun_out/prof_$T.txt
