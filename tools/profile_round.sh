#!/bin/bash
# GPU box: the per-round profile set of `bench.py` (kernel stats + the two HBM counter passes + an MFMA/SQ counter pass);
# raw output under gpurun_out/prof_$1, condensed by profiles/summarize.py into profiles/$1_*.  usage: tools/profile_round.sh r02
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- $B > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- $B > $O/sq.log 2>&1
cd $R
python3 profiles/summarize.py stats $O/trace 21 $O/${TAG}_kernel_stats.md > /dev/null
python3 profiles/summarize.py pmc $O/fetch $O/write $O/${TAG}_pmc_hbm.json > /dev/null
python3 profiles/summarize.py counters $O/sq $O/${TAG}_pmc_sq.json > /dev/null
find $O -type f ! -name "*.md" ! -name "*.json" ! -name "*.log" ! -name "*kernel_stats.csv" -delete
du -sh $O
