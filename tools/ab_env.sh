#!/bin/bash
# GPU box: same-box A/B of the training step with environment switches.  usage: tools/ab_env.sh "ENV_A=.." "ENV_B=.." [bench flags]
A="$1"; B="$2"; shift 2
for rep in 1 2; do
  for E in "$A" "$B"; do
    echo "== [$E] $*"
    env $E python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d.get('roofline',{})
print('value %.1f  ms/step %.3f  p50 %.3f | %s %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['step_ms']['p50'], r.get('kernel','?'), 1e3*r.get('avg_ms',0), r.get('frac',0)))
print(' others:', {o['kernel']: round(1e3*o['avg_ms'],1) for o in d.get('roofline_others',[])})
am=d.get('attention_module'); print(' attention_module ms_per_block', am and round(am['ms_per_block'],4))
"
  done
done
