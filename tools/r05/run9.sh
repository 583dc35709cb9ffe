#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py::test_full_size_training_trajectory_against_oracle > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
