#!/bin/bash
# usage: tools/ab_attn.sh <alt lib under tools/_bin> [rounds]  -- same-box A/B of the attention micro-benchmark (dropout 0.1, slope gradient on)
alt=$1; n=${2:-2}
for i in $(seq $n); do
  for v in new alt; do
    if [ $v = alt ]; then export SPN_LIB=tools/_bin/$alt; else unset SPN_LIB; fi
    echo "== $v"; DROP=0.1 python tools/bench_attn.py 2>/dev/null | grep causal
  done
done
