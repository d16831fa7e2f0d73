#!/bin/bash
# same-box sweep of library tuning knobs on the whole train step (ms per step; the default first and last)
run() { env "$@" python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-dp1-forced --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.2f ms/step' % ('$*', d['ms_per_step']))"; }
run X=0
for kv in SPN_GEMM_NGROUP=4 SPN_GEMM_NGROUP=16 SPN_GEMM_SPLIT_BLOCKS=512 SPN_GEMM_SPLIT_BLOCKS=128 SPN_GLU_PERSIST=1 SPN_GLU_PERSIST=4 SPN_GEMM_PERSIST=1 SPN_GEMM_PERSIST=4 SPN_GEMM_SLICE_XCD=0 SPN_LN_BWD_BLOCKS=1024 SPN_LN_BWD_BLOCKS=4096 SPN_EMBED_STATS_BLOCKS=1024 SPN_EMBED_SCATTER_BLOCKS=512 SPN_GLU_BWD_DUO=0 SPN_GEMM_DUO_NGROUP=4 SPN_GEMM_DUO_NGROUP=16 SPN_ATTN_ORDER=0; do run $kv; done
run X=0
