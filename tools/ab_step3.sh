#!/bin/bash
# usage: tools/ab_step3.sh <rounds> <alt lib> [<alt lib> ...]   -- same-box A/B of the whole train step: shipped libspn.so vs variant builds under tools/_bin
n=$1; shift
for i in $(seq $n); do
  for v in new "$@"; do
    if [ $v = new ]; then unset SPN_LIB; else export SPN_LIB=tools/_bin/$v; fi
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-dp1-forced 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; a=r['attention']['with_band']; e=r['elementwise']
print('$v', 'ms/step %.2f' % d['ms_per_step'], 'gemm %.1f' % r['gemm_ms_per_step'], 'attn %.1f' % a['ms_per_step'], 'elem %.1f' % e['ms_per_step'])"
  done
done
