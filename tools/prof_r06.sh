#!/bin/bash
# Round-6 profiling bundle (run through gpurun; every rocprofv3 invocation puts the program itself behind `--`):
#   1. the default bench line (phases, sustained, cpu_baseline, dp1_forced, decode_c5 included) -> gpurun_out/$tag/bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same train step                                  -> gpurun_out/$tag/kernel_stats_step.csv
#   3. ATen ops by call site of one step                                                        -> gpurun_out/$tag/aten_ops.txt
R=$GRAFT_REPO_ROOT
tag=${1:-r06}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
if [ "${2:-bench}" = "bench" ]; then
  python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  tail -c 300 $O/bench_default.json; echo
fi
python3 tools/aten_ops.py > $O/aten_ops.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline --no-phases --sustained-seconds 0 > /tmp/prof_step.log 2>&1
find /tmp/prof_step -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_step.csv \;
tail -2 /tmp/prof_step.log | cut -c1-300
ls -la $O
# 4. BASELINE configs 2 and 4 as throughput lines (their parity tests are tests/test_parity_c2_gpu.py and the two-rank halves of test_dp_gpu.py):
#    C2 = 64 x 1024 on one GPU; C4 shard = the same shard with the data-parallel machinery on (one-rank RCCL group, bucketed all-reduce)
if [ "${3:-}" = "configs" ]; then
  cd $R
  python3 bench.py --preset c2 --seq 1024 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --sustained-seconds 0 > $O/bench_c2_seq1024.json 2>> $O/bench_default.err
  python3 bench.py --preset c2 --seq 1024 --steps 20 --warmup 5 --no-cpu-baseline --no-decode --sustained-seconds 0 --force-dp --no-dp1-forced > $O/bench_c4_shard_seq1024_forcedp.json 2>> $O/bench_default.err
  tail -c 200 $O/bench_c2_seq1024.json; echo
fi
# 5. counters and the decode profile (own passes; rocprofv3 --pmc with --kernel-trace only)
if [ "${4:-}" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python3 $R/tools/prof_decode.py > /tmp/prof_dec.log 2>&1
  find /tmp/prof_dec -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_decode.csv \;
  DROP=0.1 REPS=5 $R/tools/pmc_run.sh ${tag}a attn tools/bench_attn.py > $O/attention_pmc.txt 2>&1
  $R/tools/pmc_step_traffic.sh ${tag}t > $O/traffic.log 2>&1
  cp $R/gpurun_out/${tag}t_step_traffic.json $O/step_traffic.json 2>/dev/null
  cp $R/gpurun_out/${tag}t_decode_traffic.json $O/decode_traffic.json 2>/dev/null
  ls -la $O
fi
