#!/usr/bin/env python3
"""HBM bytes per launch of the fused kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_record.sh.
(2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
read, WRITE_SIZE is exact (/opt/skills/guides/MI355X_MICROARCH.md, HBM).  Usage: pmc_traffic.py <dir> [bench args]"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
args = sys.argv[2:]


def opt(name, default):
    return args[args.index(name) + 1] if name in args else default


def mean_per_launch(counter):
    per = {}
    name = None
    for f in glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "eval_kernel" not in row.get("Kernel_Name", "") or row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    vals = list(per.values())
    return (sum(vals) / len(vals) if vals else None), len(vals), name


fs, nf, kname = mean_per_launch("FETCH_SIZE")
ws, nw, _ = mean_per_launch("WRITE_SIZE")
out = {"workload": opt("--workload", "mixed-6x64") + ("_resonly" if "--residual-only" in args else ""),
       "batch": int(opt("--batch", "65536")), "kernel": kname, "launches": [nf, nw],
       "FETCH_SIZE_KiB_per_launch": fs, "WRITE_SIZE_KiB_per_launch": ws}
if fs is not None and ws is not None:
    out["hbm_bytes_per_launch_raw"] = (fs + ws) * 1024
    out["hbm_bytes_per_launch"] = (2 * fs + ws) * 1024
out["correction"] = ("(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 64 B per 128-B request "
                     "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; separate --pmc passes")
out["source"] = "tools/gpu_record.sh -> rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, bench.py --steps 4 --warmup 1 --settle-ms 0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gelato_amd import _lib  # noqa: E402  (provenance only: which build these counters describe)
out.update({"build_" + k: v for k, v in _lib.build_info().items()})
print(json.dumps(out, indent=1))
