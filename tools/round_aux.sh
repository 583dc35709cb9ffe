#!/bin/bash
# GPU box: the auxiliary legs of a round's measurement set -> gpurun_out/$1_*   usage: tools/round_aux.sh r03
TAG=${1:-r03}
python bench.py --no-cpu-baseline --local-batch 8 > gpurun_out/${TAG}_bench_b8.json 2>/dev/null
python bench.py --no-cpu-baseline --rel-pos 1 > gpurun_out/${TAG}_bench_relpos.json 2>/dev/null
python bench.py --no-cpu-baseline --data host > gpurun_out/${TAG}_bench_host.json 2>/dev/null
for f in b8 relpos host; do python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_$f.json"))
print("$f", round(d["value"],1), "samples/s", round(d["ms_per_step"],2), "ms/step", {k: round(v["samples_per_s"],1) for k,v in (d.get("host_pipeline") or {}).items()})
PY
done
( python tools/run_cfg.py bench_geo_depth24_e192_invar 2 6; python tools/run_cfg.py bench_depth12_e128_2step 2 6; python tools/run_cfg.py swin_73var 1 4; python tools/run_cfg.py swin_73var_geo_depth12_chweight_invar 1 4 ) 2>/dev/null | grep "local batch" | tee gpurun_out/${TAG}_configs.txt
