#!/bin/bash
# GPU box: SQ / LDS counters of the attention-backward probe (tools/probe_attn3_stamps.py without stamps would do; the
# stamp build costs nothing measurable) -> gpurun_out/pmc_attn3/{sq,lds}.json.  One PMC pass per counter group.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn3
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSWV2_ATTN3_STAMPS -o /tmp/libswv2_a3stamps.so $R/swin_v2_weather_amd/csrc/*.hip 2>/dev/null
export SWV2_PROBE_NOBUILD=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/p -o p --output-format csv -- python3 $R/tools/probe_attn3_stamps.py > $O/p.log 2>&1
cd $R && python3 profiles/summarize.py counters $O/p $O/lds.json > /dev/null
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/q -o q --output-format csv -- python3 $R/tools/probe_attn3_stamps.py > $O/q.log 2>&1
cd $R && python3 profiles/summarize.py counters $O/q $O/sq.json > /dev/null
find $O -type f ! -name "*.json" ! -name "*.log" -delete
python3 - <<'PY'
import json
for f in ("lds","sq"):
    d=json.load(open(f"gpurun_out/pmc_attn3/{f}.json"))
    for k,v in d.items():
        if "attn_bwd" in k: print(f, k[:50], json.dumps(v)[:1500])
PY
