#!/bin/bash
# GPU box: HBM counter passes (FETCH_SIZE, WRITE_SIZE: separate runs, as the guide prescribes) of the bench command with the CPB bias
# (--rel-pos 1) -> gpurun_out/${TAG}_pmc_hbm_relpos.json (per-launch means per kernel)      usage: tools/pmc_relpos.sh r05
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_relpos_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --settle 2 --no-cpu-baseline --no-secondary --rel-pos 1"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $B > $O/write.log 2>&1
cd $R
python3 profiles/summarize.py pmc $O/fetch $O/write gpurun_out/${TAG}_pmc_hbm_relpos.json > /dev/null
find $O -type f ! -name "*.log" -delete
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmc_hbm_relpos.json"))
rows = [(k, v) for k, v in d.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v]
for k, v in sorted(rows, key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:16]:
    print(f"{k[:64]:64s} {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB per launch")
PY
