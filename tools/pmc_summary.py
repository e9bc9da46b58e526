#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for the bench kernel: per-launch averages of every counter.
Usage: pmc_summary.py <dir with pmc_*/ subdirs> [kernel substring]"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "eval_kernel"
agg = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            d = agg.setdefault(k, {})
            key = row.get("Dispatch_Id")
            d[key] = d.get(key, 0.0) + float(row["Counter_Value"])
out = {}
for k, d in sorted(agg.items()):
    vals = list(d.values())
    out[k] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    fs, ws = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads
    # (/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE is exact.
    out["hbm_bytes_per_launch_raw"] = (fs + ws) * 1024
    out["hbm_bytes_per_launch"] = (2 * fs + ws) * 1024
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
