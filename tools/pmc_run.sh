#!/bin/bash
# usage: tools/pmc_run.sh <tag> <kernel-substring> <script.py> -- SQ counter passes for the kernels whose name contains the substring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; filt=$2; script=$3
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$tag$i -o p -- python3 $R/$script > /tmp/pmc_$tag$i.log 2>&1
  f=$(find /tmp/pmc_$tag$i -name '*counter_collection.csv' | head -1)
  FILT="$filt" python3 - "$f" <<'PY'
import csv, sys, collections, os
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if os.environ["FILT"] not in r['Kernel_Name']: continue
    import re
    m = re.search(r'(\w+_kernel(<[\w, ]+>)?)', r['Kernel_Name'])
    k = m.group(1) if m else r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k)
    for c, v in d.items(): print(f"   {c:32s} {v:.4g}")
PY
done
