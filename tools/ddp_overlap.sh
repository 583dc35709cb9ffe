#!/bin/bash
# GPU box: does the gradient all-reduce run beside the backward kernels?  Kernel trace of `bench.py --force-ddp` (a one-rank RCCL
# group: the collective kernels are launched exactly as with N ranks, on the reducer's stream) -> per RCCL kernel, the compute
# kernels that ran while it was in flight.  usage: tools/ddp_overlap.sh r03 -> gpurun_out/r03_ddp_overlap.md
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ddp_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python3 $R/bench.py --force-ddp --steps 6 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
cd $R
python3 tools/ddp_overlap.py $O/trace gpurun_out/${TAG}_ddp_overlap.md
tail -2 $O/trace.log | cut -c1-300
find $O -type f ! -name "*.log" -delete
