#!/bin/bash
# HBM-side traffic of one GEMM shape (env M N K TA TB F32), separate --pmc passes as the microarch guide prescribes.
# usage: tools/pmc_traffic.sh <tag>  -> prints FETCH_SIZE / WRITE_SIZE per launch (raw counter units) and writes gpurun_out/traffic_<tag>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tr_${tag}_$c -o p -- python3 $R/tools/gemm_one.py > /tmp/tr_${tag}_$c.log 2>&1
done
python3 - "$tag" "$R" <<'PY'
import csv, sys, glob, json, os
tag, R = sys.argv[1], sys.argv[2]
out = {"shape": {k: int(os.environ.get(k, d)) for k, d in (("M", 8192), ("N", 8192), ("K", 8192), ("TA", 0), ("TB", 0), ("F32", 0))}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/tr_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c] = {"launches": len(vals), "per_launch_raw": sum(vals) / max(1, len(vals)), "min": min(vals), "max": max(vals)}
    names = {r["Kernel_Name"].split("(")[0][-60:] for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"]}
    out["kernels"] = sorted(names)
print(json.dumps(out, indent=1))
os.makedirs(f"{R}/gpurun_out", exist_ok=True)
json.dump(out, open(f"{R}/gpurun_out/traffic_{tag}.json", "w"), indent=1)
PY
