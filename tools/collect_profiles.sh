#!/bin/bash
# (here) copies what tools/record_all.sh + tools/record_others.sh left under gpurun_out/<tag>/ into profiles/<tag>/ (tracked) and refreshes
# the static traffic / fp64 files bench.py reads.  tools/collect_profiles.sh <tag>
TAG=${1:-r06}
cd "$(dirname "$0")/.."
mkdir -p profiles/$TAG/others
for w in mixed dense stress 3x32res; do
  for f in bench.json bench_profiled.json kernel_stats.csv traffic.json fp64.json; do cp gpurun_out/$TAG/$w/$f profiles/$TAG/${w}_$f; done
done
cp gpurun_out/$TAG/others/{others.json,kernel_stats.csv,pmc_FETCH_SIZE.csv,pmc_WRITE_SIZE.csv,pmc_busy.csv} profiles/$TAG/others/
# round 4 on: the chip's clock / power under load, the batch scan, the simulated-world shard step, the parity margins, B = 1, stamps, memory-system counters
for f in power_clock.json parity_margins.json callback_b1.jsonl callback_python.jsonl b1_breakdown.json b1_breakdown_compact_path.json instruction_mix.txt bench_default.json batch_scan.json aero_fused_default.json aero_fused_fused.json aero_fused_kernel_stats.csv placement_alternation.txt; do
  [ -f gpurun_out/$TAG/$f ] && cp gpurun_out/$TAG/$f profiles/$TAG/$f
done
python3 - $TAG <<'PY'
import json, shutil, sys
tag = sys.argv[1]
for w in ("mixed", "dense", "stress", "3x32res"):
    for kind in ("traffic", "fp64"):
        d = json.load(open("profiles/%s/%s_%s.json" % (tag, w, kind)))
        shutil.copy("profiles/%s/%s_%s.json" % (tag, w, kind), "profiles/%s_%s_B%d.json" % (kind, d["workload"], d["batch"]))
        print("profiles/%s_%s_B%d.json" % (kind, d["workload"], d["batch"]))
PY
