#!/bin/bash
# (here) copies what tools/record_all.sh + tools/record_others.sh left under gpurun_out/<tag>/ into profiles/<tag>/ (tracked) and refreshes
# the static traffic / fp64 files bench.py reads.  tools/collect_profiles.sh <tag>
TAG=${1:-r03}
cd "$(dirname "$0")/.."
mkdir -p profiles/$TAG/others
for w in mixed dense stress 3x32res; do
  for f in bench.json bench_profiled.json kernel_stats.csv traffic.json fp64.json; do cp gpurun_out/$TAG/$w/$f profiles/$TAG/${w}_$f; done
done
cp gpurun_out/$TAG/others/{others.json,kernel_stats.csv,pmc_FETCH_SIZE.csv,pmc_WRITE_SIZE.csv,pmc_busy.csv} profiles/$TAG/others/
python3 - $TAG <<'PY'
import json, shutil, sys
tag = sys.argv[1]
for w in ("mixed", "dense", "stress", "3x32res"):
    for kind in ("traffic", "fp64"):
        d = json.load(open("profiles/%s/%s_%s.json" % (tag, w, kind)))
        shutil.copy("profiles/%s/%s_%s.json" % (tag, w, kind), "profiles/%s_%s_B%d.json" % (kind, d["workload"], d["batch"]))
        print("profiles/%s_%s_B%d.json" % (kind, d["workload"], d["batch"]))
PY
