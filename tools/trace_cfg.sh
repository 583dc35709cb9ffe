#!/bin/bash
# GPU box: kernel trace of a few Trainer steps of a yaml config -> gpurun_out/$TAG_kernel_stats.md
# usage: tools/trace_cfg.sh tag config batch steps
TAG=$1; CFG=$2; B=${3:-2}; STEPS=${4:-6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/run_cfg.py $CFG $B $STEPS > $O/trace.log 2>&1
cd $R
python3 profiles/summarize.py stats $O/trace $((STEPS + 3)) gpurun_out/${TAG}_kernel_stats.md > /dev/null
find $O -type f ! -name "*.log" -delete
grep "local batch" $O/trace.log; head -36 gpurun_out/${TAG}_kernel_stats.md
