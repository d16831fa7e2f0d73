#!/bin/bash
# usage: tools/prof_attn_tree.sh <tag> <tree root>  -- per-kernel times (rocprofv3 kernel trace) of the attention micro-benchmark run from
# a tree (the working tree, or an older one under tools/_bin/); env (RAGGED, FULLMASK, DROP, ...) passes through to tools/bench_attn.py
tag=$1; R=$(cd ${2:-.} && pwd)
cd /tmp && export TMPDIR=/tmp
export DROP=${DROP:-0.1} REPS=${REPS:-5}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn_$tag -o a -- python3 $R/tools/bench_attn.py > /tmp/prof_attn_$tag.log 2>&1
f=$(find /tmp/prof_attn_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
