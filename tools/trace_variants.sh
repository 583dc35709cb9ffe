#!/bin/bash
# GPU box: per-window cycle traces of prebuilt stamped variants (tools/build_stamped.sh).  usage: tools/trace_variants.sh st_a.so st_b.so ...
first=1
for rep in 1 2; do
for SO in "$@"; do
  echo "== [$SO] rep $rep"
  if [ $first = 1 ]; then PROBE_SO=$PWD/tools/_so/$SO python tools/probe_attn_bwd_windows.py 2>&1 | grep -v amdgpu.ids | grep -A1 "two-phase\|streamed" | grep -v "^--"; first=0
  else PROBE_ONLY_STREAMED=1 PROBE_SO=$PWD/tools/_so/$SO python tools/probe_attn_bwd_windows.py 2>&1 | grep -v amdgpu.ids | grep -A1 "streamed" | grep -v "^--"; fi
done
done
