#!/bin/bash
# kernels other than the fused one on the record: bench-style JSON, rocprofv3 --kernel-trace --stats summary, FETCH_SIZE / WRITE_SIZE
# passes -> gpurun_out/<tag>/others/{others.json, kernel_stats.csv, pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv}.  tools/record_others.sh <tag> [workload]
TAG=${1:-r03}; WL=${2:-mixed-6x64}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG/others; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 python3 $R/tools/other_kernels.py $WL > $OUT/others.json 2> $OUT/others.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/tools/other_kernels.py $WL > $OUT/others_profiled.json 2> $OUT/prof.err
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  OK_REPS=4 timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 $R/tools/other_kernels.py $WL > /dev/null 2> $OUT/pmc_$c.err
  python3 - $OUT/pmc_$c $c > $OUT/pmc_$c.csv <<'PY'
import csv, glob, os, sys
root, counter = sys.argv[1], sys.argv[2]
per = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != counter:
            continue
        k = (row["Kernel_Name"].split("(")[0], row["Dispatch_Id"], row.get("Grid_Size", ""))
        per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
agg = {}
for (name, _, grid), v in per.items():
    agg.setdefault((name, grid), []).append(v)
print("kernel,grid_size,dispatches,%s_KiB_mean_per_dispatch" % counter)
for (name, grid), vs in sorted(agg.items()):
    print('"%s",%s,%d,%.1f' % (name, grid, len(vs), sum(vs) / len(vs)))
PY
done
# datapath occupancy per kernel (what binds the ones that are far from the HBM roofline)
OK_REPS=4 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_busy -o pmc -- python3 $R/tools/other_kernels.py $WL > /dev/null 2> $OUT/pmc_busy.err
python3 - $OUT/pmc_busy > $OUT/pmc_busy.csv <<'PY'
import csv, glob, os, sys
per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0], row.get("Grid_Size", ""), row["Dispatch_Id"])
        per.setdefault(k, {}).setdefault(row["Counter_Name"], 0.0)
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
agg = {}
for (name, grid, _), c in per.items():
    a = agg.setdefault((name, grid), {"n": 0})
    a["n"] += 1
    for k, v in c.items(): a[k] = a.get(k, 0.0) + v
print("kernel,grid_size,dispatches,valu_insts_per_wave,valu_busy,wait_inst_share")
for (name, grid), a in sorted(agg.items()):
    if not a.get("SQ_WAVES") or not a.get("GRBM_GUI_ACTIVE"): continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    print('"%s",%s,%d,%.1f,%.3f,%.3f' % (name, grid, a["n"], a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                                       a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0)))
PY
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_busy
echo "== others"; cat $OUT/others.json; head -12 $OUT/kernel_stats.csv; cat $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_busy.csv
