#!/bin/bash
# usage: tools/ab_decode.sh [rounds]  -- same-box A/B of the C5 greedy decode (tools/bench_dec_pair.py: five launches per layer pair and
# the persistent launch, L = 4096): the tree at the start of the round (tools/_bin/tree_base) against the working tree, alternating
n=${1:-2}
here=$(pwd)
mkdir -p tools/_bin/tree_base/tools && cp tools/bench_dec_pair.py tools/_bin/tree_base/tools/
for i in $(seq $n); do
  (cd tools/_bin/tree_base && TIMELINE=0 python tools/bench_dec_pair.py 4096 2>/dev/null | grep "us per note" | sed 's/^/base  /')
  (cd $here && TIMELINE=0 python tools/bench_dec_pair.py 4096 2>/dev/null | grep "us per note" | sed 's/^/new   /')
done
