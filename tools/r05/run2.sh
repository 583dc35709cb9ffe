#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cpb_multi or local_batch_8" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
python tools/probe_host_time.py --profile 1 > $O/host_nopos.txt 2>&1; head -8 $O/host_nopos.txt
python tools/probe_host_time.py --rel-pos 1 > $O/host_relpos.txt 2>&1; head -8 $O/host_relpos.txt
