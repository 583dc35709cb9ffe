#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
pick='import json,sys
d=json.loads([l for l in sys.stdin if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"],1), "samples/s", d["step_ms"]["sequence"])'
B="python bench.py --no-cpu-baseline --no-secondary"
for i in 1 2 3; do $B 2>/dev/null | python -c "$pick" default; done
for i in 1 2; do $B --no-kernel-timing 2>/dev/null | python -c "$pick" no_kernel_timing; done
for i in 1 2; do $B --steps 20 2>/dev/null | python -c "$pick" steps20; done
