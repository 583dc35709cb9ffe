#!/bin/bash
# GPU box: kernel-trace of bench.py at local batch 1 / 2 / 4 -> per-kernel average durations; tools/batch_fit.py fits
# t(B) = a + b B per kernel (a = batch-independent cost of a launch: ramp, drain, quantisation; b = per-sample cost)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/batch_fit
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 2 4; do
  rocprofv3 --kernel-trace --stats -d $O/b$B -o t --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --local-batch $B > $O/b$B.log 2>&1
  find $O/b$B -type f ! -name "*kernel_stats.csv" -delete
done
cd $R && python3 tools/batch_fit.py $O
