#!/bin/bash
# GPU box: the per-round profile set of `bench.py`: kernel stats, the two HBM counter passes, an MFMA / SQ counter pass, the wave
# wait / active pass, and a consistency check of the three summaries bench.py's `roofline` rests on.  Raw output under
# gpurun_out/prof_$1, condensed by profiles/summarize.py into gpurun_out/prof_$1/$1_* (copy those into profiles/).
# usage: tools/profile_round.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- $B > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- $B > $O/sq.log 2>&1
cd $R
python3 profiles/summarize.py stats $O/trace 24 $O/${TAG}_kernel_stats.md > /dev/null
python3 profiles/summarize.py pmc $O/fetch $O/write $O/${TAG}_pmc_hbm.json > /dev/null
python3 profiles/summarize.py counters $O/sq $O/${TAG}_pmc_sq.json > /dev/null
python3 profiles/summarize.py mfma $O/${TAG}_pmc_sq.json $O/${TAG}_pmc_mfma.json > $O/mfma.txt
python3 profiles/summarize.py check $O/${TAG}_kernel_stats.md $O/${TAG}_pmc_hbm.json $O/${TAG}_pmc_mfma.json | tee $O/check.txt
find $O -type f ! -name "*.md" ! -name "*.json" ! -name "*.log" ! -name "*.txt" -delete
head -30 $O/${TAG}_kernel_stats.md; head -12 $O/mfma.txt
