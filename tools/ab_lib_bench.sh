#!/bin/bash
# GPU box: same-box, alternating A/B of library variants IN SITU (bench.py's training step).  usage: tools/ab_lib_bench.sh libA.so libB.so ... [-- bench flags]   ("-" = the shipped library)
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2; do
  for SO in "${LIBS[@]}"; do
    echo "== [$SO] $*"
    if [ "$SO" = "-" ]; then unset SWV2_LIB; else export SWV2_LIB=$PWD/$SO; fi
    python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d.get('roofline',{})
print('value %.1f  ms/step %.3f  p50 %.3f | %s %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['step_ms']['p50'], r.get('kernel','?'), 1e3*r.get('avg_ms',0), r.get('frac',0)))
"
  done
done
