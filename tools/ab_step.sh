#!/bin/bash
# usage: tools/ab_step.sh <alt lib under tools/_bin> [rounds]   -- same-box A/B of the whole train step: shipped libspn.so vs a variant build
alt=$1; n=${2:-2}
for i in $(seq $n); do
  for v in new alt; do
    if [ $v = alt ]; then export SPN_LIB=tools/_bin/$alt; else unset SPN_LIB; fi
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-dp1-forced 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=r['attention']['with_band']; e=r['elementwise']
print('$v', 'ms/step %.1f' % d['ms_per_step'], 'gemm %.1f' % r['gemm_ms_per_step'], 'attn %.1f (fwd %.1f bwd %.1f)' % (a['ms_per_step'], a['fwd_ms_per_step'], a['bwd_ms_per_step']), 'elem %.1f' % e['ms_per_step'])"
  done
done
