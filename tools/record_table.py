#!/usr/bin/env python3
"""(here) The per-workload table of DESIGN.md 6 / profiles/README.md from the files of one recording (profiles/<tag>/).

Usage:  python tools/record_table.py [tag]        (default r04)"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"mixed": "mixed-6×64 (65536)", "dense": "dense-6×64 (65536)", "stress": "stress-12×128 (16384)",
         "3x32res": "3×32 residual-only (65536)"}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    d = os.path.join(REPO, "profiles", tag)
    load = lambda n: json.load(open(os.path.join(d, n)))
    print("| workload (batch) | `value` evals/s | `frac` (HBM, `A_min`) | kernel ms: HIP events / rocprof avg | traffic / `A_min` | "
          "fp64 datapath busy (profiled pass) | wait_inst | VALU / wave | CPU oracle, 1 thread |")
    print("|---|---|---|---|---|---|---|---|---|")
    shas = set()
    for w, name in NAMES.items():
        b, t, f = load(w + "_bench.json"), load(w + "_traffic.json"), load(w + "_fp64.json")
        shas.add((t["build_so_sha256"][:12], (t.get("build_git_head") or "")[:7]))
        with open(os.path.join(d, w + "_kernel_stats.csv")) as fh:
            rows = [r for r in csv.DictReader(fh) if "eval_kernel" in r["Name"]]
        avg = float(rows[0]["AverageNs"]) / 1e6
        r = b["roofline"]
        print("| %s | %.3g M | %.3f | %.3f / %.3f | %.3f | %.2f (VALU %.2f + MFMA %.2f) | %.2f | %s | %.3g |" % (
            name, b["value"] / 1e6, r["frac"], r["kernel_ms"], avg, t["hbm_bytes_per_launch"] / r["algorithmic_bytes_per_launch"],
            f["fp64_pipe_busy"], f["valu_busy"], f["mfma_busy"], f["wait_inst_share"], format(round(f["valu_instructions_per_wave"]), ","),
            b["cpu_baseline"]["value"]))
    bd = load("bench_default.json")
    print("\nbench_default.json: %.2f M evals/s, frac %.3f, kernel %.3f ms" % (bd["value"] / 1e6, bd["roofline"]["frac"], bd["roofline"]["kernel_ms"]))
    print("builds on record:", sorted(shas))


if __name__ == "__main__":
    main()
