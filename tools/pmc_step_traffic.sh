#!/bin/bash
# HBM-side traffic per launch of the memory-bound kernels of the train step (and of the decode launch): two separate rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "Per-pass counter slots"), the program itself behind
# `--`, counter collection restricted to the named kernels (a whole step under per-dispatch counters takes > 10 minutes).
# usage (through gpurun): tools/pmc_step_traffic.sh <tag>   -> gpurun_out/<tag>_step_traffic.json, gpurun_out/<tag>_decode_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
INC='ln_bwd|ln_fwd_kernel|adaln_fwd_kernel|embed_bwd|embed_fwd|attn_fwd_kernel|attn_bwd|gemm_duo8_glu_bwd|gemm_pp_kernel<false, false, unsigned short, 1|adamw_kernel|seg_sum|ce_fwd'
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$INC" --output-format csv -d /tmp/st_${tag}_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline --no-phases --sustained-seconds 0 > /tmp/st_${tag}_$c.log 2>&1
  L=384 timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "dec_pair_kernel|dec_head|dec_step_begin" --output-format csv -d /tmp/sd_${tag}_$c -o p -- python3 $R/tools/prof_decode.py > /tmp/sd_${tag}_$c.log 2>&1
done
python3 $R/tools/pmc_step_traffic.py /tmp/st_${tag} $R/gpurun_out/${tag}_step_traffic.json
python3 $R/tools/pmc_step_traffic.py /tmp/sd_${tag} $R/gpurun_out/${tag}_decode_traffic.json
