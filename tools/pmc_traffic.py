#!/usr/bin/env python3
"""HBM bytes per launch of the fused kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_record.sh.
(2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
read, WRITE_SIZE is exact (/opt/skills/guides/MI355X_MICROARCH.md, HBM).  Usage: pmc_traffic.py <dir> [bench args]"""
import csv
import glob
import json
import os
import sys


def launch_policy(workload, batch, resonly):
    """what the HOST side decided for this launch (instantiation, wavefronts, LDS order, store policy: gel_launch_info + the
    environment switches that change it) -- recorded with the counters, so that bench.py does not report them for a library whose
    device code is the same but whose launch policy is not"""
    from gelato_amd import Engine, con_dynamics, problem
    pd, ud, _c, _x = problem.make_problem(workload)
    E = Engine(con_dynamics.problem_arrays(pd, ud))
    return {"launch_info": E.launch_info(batch, True, not resonly), "num_chunks": E.num_chunks()}



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
try:
    out["launch_policy"] = launch_policy(out["workload"].replace("_resonly", ""), out["batch"], out["workload"].endswith("_resonly"))
except Exception as ex:  # noqa: BLE001
    out["launch_policy"] = {"error": str(ex)[:200]}
print(json.dumps(out, indent=1))
