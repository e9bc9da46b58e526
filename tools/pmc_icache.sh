#!/bin/bash
# instruction-cache and instruction-fetch counters of the fused kernel (GPU box): tools/pmc_icache.sh "<bench args>"
R=$GRAFT_REPO_ROOT; BA="$1"
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL_[A-Z_]*" | sort -u | tr '\n' ' '; echo
for grp in "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ"; do
  OUT=/tmp/pmcic; rm -rf $OUT
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BA > /dev/null 2> $OUT.err || { echo "group failed: $grp"; tail -2 $OUT.err; continue; }
  python3 - $OUT <<'PY'
import csv, glob, sys
per = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            d = per.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)[-4:]
agg = {}
for i in ids:
    for k, v in per[i].items(): agg[k] = agg.get(k, 0.0) + v / len(ids)
w = agg.get("SQ_WAVES", 1.0)
for k in sorted(agg):
    if k != "SQ_WAVES": print("  %-32s %14.1f per wavefront   (%.4g per launch)" % (k, agg[k] / w, agg[k]))
PY
done
