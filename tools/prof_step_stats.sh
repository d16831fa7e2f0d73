#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default train step (the program itself behind `--`) -> gpurun_out/<tag>/kernel_stats_step.csv
R=$GRAFT_REPO_ROOT; tag=${1:-r05s}
mkdir -p $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline > /tmp/prof_step.log 2>&1
find /tmp/prof_step -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats_step.csv \;
head -5 $R/gpurun_out/$tag/kernel_stats_step.csv | cut -c1-160
