#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cpb or stage_level" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log | grep -E "passed|failed|^E " 
python tools/probe_cpb_multi.py 2>&1 | grep -v amdgpu
