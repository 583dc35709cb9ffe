#!/bin/bash
# GPU box: LDS / issue counters of the wide-GEMM probe for a build variant: tools/pmc_wide.sh "<macros>" [shape index]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_wide
rm -rf $O; mkdir -p $O
SO=/tmp/libswv2_pmc.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $1 -o $SO $R/swin_v2_weather_amd/csrc/*.hip 2>/dev/null
export SWV2_LIB=$SO ONLY=${2:-2}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/probe_wide_gemm.py | grep "K="
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT -d $O/lds -o p --output-format csv -- python3 $R/tools/probe_wide_gemm.py > $O/lds.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- python3 $R/tools/probe_wide_gemm.py > $O/sq.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_wide"
for sub in ("lds", "sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_nt" not in k and "Cijk" not in k: continue
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(sub, k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions) n=", len(next(iter(d.values()))))
PY
find $O -type f ! -name "*.log" -delete
