#!/bin/bash
# dynamic instruction mix of the bench kernel (per wavefront): tools/pmc_mix.sh <tag> [bench args]   (GPU box; two --pmc passes)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/mix_a -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/mix_a.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/mix_b -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/mix_b.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/mix_c -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/mix_c.err
python3 - $OUT <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
tot = {}
for sub in ("mix_a", "mix_b", "mix_c"):
    per = {}
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" not in r.get("Kernel_Name", ""): continue
            per.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per, key=int)[-4:]
    agg = {}
    for i in ids:
        for k, v in per[i].items(): agg[k] = agg.get(k, 0.0) + v / len(ids)
    w = agg.get("SQ_WAVES", 1.0)
    for k, v in sorted(agg.items()):
        if k != "SQ_WAVES": tot[k] = v / w
    tot["SQ_WAVES"] = w
print(" ".join("%s=%.1f" % (k, v) for k, v in sorted(tot.items())))
PY
rm -rf $OUT/mix_a $OUT/mix_b $OUT/mix_c
