#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5r
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python3 $R/tools/bench_render.py --module-notes 40 > /tmp/prof_r.log 2>&1
find /tmp/prof_r -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r5r/render_kernel_stats.csv \;
tail -3 /tmp/prof_r.log | cut -c1-300
