# ablation builds (tools/build_variants.sh) under rocprofv3 --pmc: VALU / store instructions per wavefront and kernel time.  VARIANTS="base novel ..." tools/ablation_counts.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in ${VARIANTS:-base novel nopos noqm lean nodx nostore}; do
  OUT=gpurun_out/abl/$n; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp; GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_$n.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras --workload dense-6x64 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/err )
  python3 - $OUT $n <<'PY'
import csv, glob, os, sys
root, name = sys.argv[1], sys.argv[2]
per, dur = {}, {}
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" not in r.get("Kernel_Name", ""): continue
        per.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
ids = sorted(per, key=int)[-4:]
a = {}
for i in ids:
    for k, v in per[i].items(): a[k] = a.get(k, 0.0) + v / len(ids)
w = a["SQ_WAVES"]; cyc = a["GRBM_GUI_ACTIVE"] / 8
print("%-8s valu/wave %7.1f salu %6.1f lds %5.1f vmem_wr %5.1f  ms %.3f  valu_busy %.3f mfma_busy %.3f" % (
    name, a["SQ_INSTS_VALU"] / w, a["SQ_INSTS_SALU"] / w, a["SQ_INSTS_LDS"] / w, a["SQ_INSTS_VMEM_WR"] / w,
    sum(dur[i] for i in ids if i in dur) / max(1, len([i for i in ids if i in dur])) / 1e6, a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
PY
  rm -rf $OUT
done
