#!/bin/bash
# VALU / SALU instructions per wavefront of the fused kernel for one library (GPU box): tools/pmc_valu_variant.sh <lib.so or ""> "<bench args>"
R=$GRAFT_REPO_ROOT; LIBV="$1"; BA="$2"
export TMPDIR=/tmp; cd /tmp; OUT=/tmp/pmcv; rm -rf $OUT
[ -n "$LIBV" ] && export GELATO_AMD_LIB=$R/$LIBV
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras --no-other-configs $BA > /dev/null 2> $OUT.err
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
print("  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), agg[k] / w) for k in sorted(agg) if k != "SQ_WAVES"))
PY
