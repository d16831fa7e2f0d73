#!/bin/bash
# usage: tools/prof_attn.sh <tag> [alt lib]  -- per-kernel times of the attention micro-benchmark (rocprofv3 kernel trace), dropout 0.1
tag=$1; alt=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
if [ -n "$alt" ]; then export SPN_LIB=$R/tools/_bin/$alt; fi
export DROP=0.1 REPS=5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn_$tag -o a -- python3 $R/tools/bench_attn.py > /tmp/prof_attn_$tag.log 2>&1
f=$(find /tmp/prof_attn_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn' in r['Name']:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
