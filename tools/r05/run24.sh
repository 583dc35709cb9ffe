#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in new vr vm vp new; do
  if [ $v = new ]; then unset SWV2_LIB; else export SWV2_LIB=$R/tools/r05/_so/libswv2_$v.so; fi
  echo "== $v"; python -m pytest tests/test_gpu_parity.py -x -q -s -k "ddp_two_ranks" 2>&1 | grep -E "elements off|FAILED|passed" | cut -c1-200
done
