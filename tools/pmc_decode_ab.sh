#!/bin/bash
# SQ instruction counters of dec_pair_kernel over one C5 greedy render (4095 notes): the round-start tree (tools/_bin/tree_base) and the
# working tree -- rocprofv3 --pmc (own pass, kernel trace only), the program itself behind `--`.  usage: tools/pmc_decode_ab.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5p}
mkdir -p $O $R/tools/_bin/tree_base/tools
cp $R/tools/prof_decode.py $R/tools/_bin/tree_base/tools/
cd /tmp && export TMPDIR=/tmp
for which in base new; do
  if [ $which = base ]; then S=$R/tools/_bin/tree_base/tools/prof_decode.py; else S=$R/tools/prof_decode.py; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmcd_$which -o p -- python3 $S > /tmp/pmcd_$which.log 2>&1
  f=$(find /tmp/pmcd_$which -name '*counter_collection.csv' | head -1)
  python3 - "$f" $which <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "dec_pair_kernel" not in r["Kernel_Name"]: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
    n += r["Counter_Name"] == "SQ_INSTS_VALU"
print(sys.argv[2], "dec_pair_kernel launches", n, "  per NOTE (4095 notes):", "  ".join(f"{k} {v / 4095:.4g}" for k, v in sorted(agg.items())))
PY
done | tee $O/decode_pmc_ab.txt
