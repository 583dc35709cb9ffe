#!/bin/bash
# GPU box: HBM counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, as the guide prescribes) of a few Trainer steps of a yaml config
# -> gpurun_out/${TAG}_pmc_hbm.json (per-launch means per kernel, units and gfx950 corrections applied by profiles/summarize.py)
# usage: tools/pmc_cfg.sh tag config batch steps
TAG=$1; CFG=$2; B=${3:-2}; STEPS=${4:-4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/tools/run_cfg.py $CFG $B $STEPS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/tools/run_cfg.py $CFG $B $STEPS > $O/write.log 2>&1
cd $R
python3 profiles/summarize.py pmc $O/fetch $O/write gpurun_out/${TAG}_pmc_hbm.json > /dev/null
find $O -type f ! -name "*.log" -delete
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmc_hbm.json"))
rows = [(k, v) for k, v in d.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v]
for k, v in sorted(rows, key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
    print(f"{k[:64]:64s} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB per launch")
PY
