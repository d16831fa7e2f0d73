#!/bin/bash
# One profiling bundle per round (run through gpurun; every rocprofv3 invocation puts the program itself behind `--`):
#   kernel-trace stats of the train step and of the C5 decode, SQ counter passes of three GEMM shapes and of the attention
#   micro-benchmark, HBM traffic of the top GEMM launch.  Results land in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
# 1. train step
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline > /tmp/prof_step.log 2>&1
find /tmp/prof_step -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats_step.csv \;
tail -1 /tmp/prof_step.log | cut -c1-200
# 2. decode
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python3 $R/tools/prof_decode.py > /tmp/prof_dec.log 2>&1
find /tmp/prof_dec -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats_decode.csv \;
# 3. GEMM counters
{
  echo "# SQ counters (rocprofv3 --pmc, separate passes), 5 launches each; MFMA-busy / busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES"
  for shape in "131072 4096 512 0 0 0" "131072 512 4096 0 1 0" "4096 512 131072 1 1 1" "8192 8192 8192 0 0 0"; do
    set -- $shape
    echo "== M=$1 N=$2 K=$3 TA=$4 TB=$5 F32=$6"
    M=$1 N=$2 K=$3 TA=$4 TB=$5 F32=$6 $R/tools/pmc_gemm.sh ${tag}g
  done
  echo "== gated forward M=131072 I=2048 K=512 (spn_gemm_glu, dropout 0.1)"
  M=131072 N=2048 K=512 GLU=1 $R/tools/pmc_gemm.sh ${tag}glu
} > $R/gpurun_out/${tag}_gemm_pmc.txt 2>&1
# 4. traffic of the top launch
M=131072 N=2048 K=512 GLU=1 $R/tools/pmc_traffic.sh ${tag}glu > /tmp/traffic.log 2>&1
