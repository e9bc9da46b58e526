#!/bin/bash
# dynamic instruction mix of the fused kernel per wavefront (one rocprofv3 --pmc pass; GPU box): tools/pmc_instmix.sh "<bench args>"
R=$GRAFT_REPO_ROOT; BA="$1"
export TMPDIR=/tmp; cd /tmp; OUT=/tmp/pmci; rm -rf $OUT
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BA > /dev/null 2> $OUT.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d ${OUT}2 -o p -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BA > /dev/null 2> ${OUT}2.err
python3 - $OUT ${OUT}2 <<'PY'
import csv, glob, sys
for root in sys.argv[1:]:
    per = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
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
        if k != "SQ_WAVES": print("  %-28s %9.1f per wavefront" % (k, agg[k] / w))
PY
