#!/bin/bash
# usage: tools/ab_env.sh "VAR=val ..." [rounds]  -- same-box A/B of the whole train step: default environment vs the given settings
n=${2:-2}
for i in $(seq $n); do
  for v in base alt; do
    if [ $v = alt ]; then pre="env $1"; else pre=""; fi
    $pre python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-dp1-forced 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=r['attention']['with_band']; e=r['elementwise']
print('$v', 'ms/step %.1f' % d['ms_per_step'], 'gemm %.1f (%.0f TF/s)' % (r['gemm_ms_per_step'], r['achieved']), 'attn %.1f' % a['ms_per_step'], 'elem %.1f' % e['ms_per_step'])"
  done
done
