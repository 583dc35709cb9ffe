#!/bin/bash
# GPU box: LDS counters of every kernel of bench.py (one PMC pass of its own) -> gpurun_out/pmc_lds/lds.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lds
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/p -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/p.log 2>&1
cd $R && python3 profiles/summarize.py counters $O/p $O/lds.json > /dev/null
find $O -type f ! -name "*.json" ! -name "*.log" -delete
