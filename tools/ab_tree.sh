#!/bin/bash
# usage: [TREE=tools/_bin/tree_base] [ARGS="--ragged"] tools/ab_tree.sh [rounds]  -- same-box A/B of the whole train step: an older tree
# (default: the one at the start of the round, tools/_bin/tree_base: `git archive <commit> scoreperformer_amd bench.py oracle include` +
# the library built from it) against the working tree
n=${1:-2}
tree=${TREE:-tools/_bin/tree_base}
here=$(pwd)
fmt='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; a=r["attention"]["with_band"]; e=r["elementwise"]
print(sys.argv[1], "ms/step %.1f" % d["ms_per_step"], "gemm %.1f (%.0f TF/s)" % (r["gemm_ms_per_step"], r["achieved"]), "attn %.1f" % a["ms_per_step"], "elem %.1f" % e["ms_per_step"])'
for i in $(seq $n); do
  (cd $tree && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode $BASE_ARGS $ARGS 2>/dev/null | python -c "$fmt" base)
  (cd $here && python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-dp1-forced --no-phases --sustained-seconds 0 $ARGS 2>/dev/null | python -c "$fmt" new)
done
