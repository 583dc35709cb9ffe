#!/bin/bash
# GPU box: kernel trace of bench.py -> gpurun_out/$1_kernel_stats.md   usage: tools/trace_bench.sh tag [bench args...]
TAG=${1:-trace}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/trace.log 2>&1
cd $R
python3 profiles/summarize.py stats $O/trace 24 gpurun_out/${TAG}_kernel_stats.md > /dev/null
find $O -type f ! -name "*.md" ! -name "*.json" ! -name "*.log" -delete
head -45 gpurun_out/${TAG}_kernel_stats.md
