#!/bin/bash
# dynamic instruction mix per wavefront of the kernels besides the fused one (tools/other_kernels.py under two rocprofv3 --pmc passes; GPU box)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; OUT=/tmp/pmcio; rm -rf $OUT ${OUT}2
OK_REPS=4 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT -o p -- python3 $R/tools/other_kernels.py mixed-6x64 > /dev/null 2> $OUT.err
OK_REPS=4 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d ${OUT}2 -o p -- python3 $R/tools/other_kernels.py mixed-6x64 > /dev/null 2> ${OUT}2.err
python3 - $OUT ${OUT}2 <<'PY'
import csv, glob, sys
for root in sys.argv[1:]:
    per = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][:40], r.get("Grid_Size", ""))
            d = per.setdefault(k, {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, agg in sorted(per.items()):
        w = agg.get("SQ_WAVES", 0.0)
        if w <= 0: continue
        print(k, "waves %.0f" % w, {c: round(v / w, 1) for c, v in sorted(agg.items()) if c != "SQ_WAVES"})
PY
