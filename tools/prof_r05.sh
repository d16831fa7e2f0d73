#!/bin/bash
# Round-5 profiling bundle (run through gpurun; every rocprofv3 invocation puts the program itself behind `--`):
#   1. the default bench line (cpu_baseline, dp1_forced, decode_c5 included)          -> gpurun_out/r05/bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same train step                         -> gpurun_out/r05/kernel_stats_step.csv
#   3. the same of the C5 decode                                                       -> gpurun_out/r05/kernel_stats_decode.csv
#   4. SQ counter passes of the attention micro-benchmark (dropout 0.1)                -> gpurun_out/r05/attention_pmc.txt
#   5. FETCH_SIZE / WRITE_SIZE passes of the HBM-bound kernels of the step and decode  -> gpurun_out/r05/step_traffic.json, decode_traffic.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline > /tmp/prof_step.log 2>&1
find /tmp/prof_step -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_step.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python3 $R/tools/prof_decode.py > /tmp/prof_dec.log 2>&1
find /tmp/prof_dec -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_decode.csv \;
DROP=0.1 REPS=5 $R/tools/pmc_run.sh r05a attn tools/bench_attn.py > $O/attention_pmc.txt 2>&1
$R/tools/pmc_step_traffic.sh r05t > $O/traffic.log 2>&1
cp $R/gpurun_out/r05t_step_traffic.json $O/step_traffic.json 2>/dev/null
cp $R/gpurun_out/r05t_decode_traffic.json $O/decode_traffic.json 2>/dev/null
ls -la $O
