#!/bin/bash
# usage: tools/pmc_gemm.sh <tag>   (env M N K ... SPN_GEMM_VARIANT forwarded)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$tag$i -o p -- python3 $R/tools/gemm_one.py > /tmp/pmc_$tag$i.log 2>&1
  f=$(find /tmp/pmc_$tag$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_' not in r['Kernel_Name']: continue
    agg[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print(f"   {c:32s} {v:.4g}")
PY
done
