#!/bin/bash
# GPU box: same-box A/B of library variants on a probe.  usage: tools/ab_lib.sh "<probe command>" libA.so libB.so ...   ("" = the shipped library)
PROBE="$1"; shift
for rep in 1 2; do
  for SO in "$@"; do
    echo "== [${SO:-shipped}]"
    if [ -n "$SO" ]; then SWV2_LIB=$PWD/$SO $PROBE 2>&1 | grep -v amdgpu.ids; else $PROBE 2>&1 | grep -v amdgpu.ids; fi
  done
done
