#!/bin/bash
# GPU box: counter passes over tools/attn_pmc.py; results under gpurun_out/pmc_attn/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_attn
mkdir -p $O
true
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/tools/attn_pmc.py > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/p1 -o p --output-format csv -- python3 $R/tools/attn_pmc.py > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM -d $O/p2 -o p --output-format csv -- python3 $R/tools/attn_pmc.py > $O/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAVE32_INSTS SQ_INSTS_SMEM -d $O/p3 -o p --output-format csv -- python3 $R/tools/attn_pmc.py > $O/p3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/p4 -o p --output-format csv -- python3 $R/tools/attn_pmc.py > $O/p4.log 2>&1
ls -R $O | head -50
# keep the merged output small: drop everything except csv files
find $O -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.log" -delete
du -sh $O
