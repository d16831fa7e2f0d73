"""Aggregates the two rocprofv3 --pmc passes of tools/pmc_step_traffic.sh per kernel name: launches, FETCH_SIZE / WRITE_SIZE per launch.
Raw counter units are KiB-like 'KB' of 1024 bytes as rocprofv3 prints them; the gfx950 correction of MI355X_MICROARCH.md (wide coalesced
streaming reads are tallied at half their bytes) is applied to FETCH_SIZE as `fetch_bytes_x2`; WRITE_SIZE is taken as printed
(calibrated here on the gated GEMM: equals the algorithmic u + g bytes).  Infinity-Cache hits are counted by both."""
import collections
import csv
import glob
import json
import re
import sys

prefix, out_path = sys.argv[1], sys.argv[2]
agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{prefix}_{c}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != c:
            continue
        name = r["Kernel_Name"]
        m = re.search(r"((?:\w+::)*\w+(?:<[^(]*>)?)\(", name)
        key = (m.group(1) if m else name)[-110:]
        per[key].append(float(r["Counter_Value"]))
    for k, v in per.items():
        a = agg.setdefault(k, {"launches": len(v)})
        a[c + "_raw_per_launch"] = sum(v) / len(v)
rows = []
for k, a in agg.items():
    f, w = a.get("FETCH_SIZE_raw_per_launch", 0.0), a.get("WRITE_SIZE_raw_per_launch", 0.0)
    a["fetch_bytes_x2"] = 2.0 * f * 1024.0
    a["write_bytes"] = w * 1024.0
    a["hbm_bytes_per_launch"] = a["fetch_bytes_x2"] + a["write_bytes"]
    rows.append((a["hbm_bytes_per_launch"] * a["launches"], k))
rows.sort(reverse=True)
out = {"note": __doc__, "kernels": {k: agg[k] for _, k in rows}}
json.dump(out, open(out_path, "w"), indent=1)
for tot, k in rows[:30]:
    a = agg[k]
    print(f"{a['launches']:5d} x {a['hbm_bytes_per_launch'] / 1e6:10.1f} MB  (fetch x2 {a['fetch_bytes_x2'] / 1e6:9.1f}, write {a['write_bytes'] / 1e6:9.1f})  {k}")
