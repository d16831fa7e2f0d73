#!/bin/bash
# usage: tools/prof_step.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-decode --no-roofline "$@" > /tmp/prof_$tag.log 2>&1
tail -1 /tmp/prof_$tag.log | cut -c1-300
mkdir -p $R/gpurun_out/prof_$tag
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_$tag/${tag}_kernel_stats.csv \;
