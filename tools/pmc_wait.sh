#!/bin/bash
# GPU box: where do the waves of every bench.py kernel spend their cycles?  SQ wait / active counters, per-launch means.
# usage: tools/pmc_wait.sh [tag]   -> gpurun_out/<tag>_pmc_wait.json
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcw_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --settle 2 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/a -o p --output-format csv -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $O/b -o p --output-format csv -- $B > $O/b.log 2>&1
cd $R
python3 profiles/summarize.py counters $O gpurun_out/${TAG}_pmc_wait.json > /dev/null
tail -3 $O/a.log $O/b.log
find $O -type f ! -name "*.log" -delete
python3 - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_pmc_wait.json"))
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CYCLES",0))[:14]:
    wc=v.get("SQ_WAVE_CYCLES",0) or 1
    print(f"{k[:52]:52s} busy/32={v.get('SQ_BUSY_CYCLES',0)/32:9.0f}  wait_any={v.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={v.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active={v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu={v.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} lds={v.get('SQ_ACTIVE_INST_LDS',0)/wc:.2f} vmem={v.get('SQ_ACTIVE_INST_VMEM',0)/wc:.2f} wait_lds={v.get('SQ_WAIT_INST_LDS',0)/wc:.2f}")
PY
