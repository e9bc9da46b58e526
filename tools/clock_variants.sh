#!/bin/bash
# effective shader clock of engine variants: GRBM_GUI_ACTIVE / 8 / kernel duration per dispatch, from ONE rocprofv3 run
# with --kernel-trace --pmc (settled: bench defaults).  Usage: clock_variants.sh "<bench args>" name...
R=$GRAFT_REPO_ROOT; BA="$1"; shift
export TMPDIR=/tmp; cd /tmp
for n in "$@"; do
  OUT=/tmp/clk_$n; rm -rf $OUT
  LIB=$R/build/variants/libgel_$n.so; [ "$n" = main ] && LIB=$R/gelato_amd/libgelato_amd.so
  GELATO_AMD_LIB=$LIB timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o c -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $BA > /dev/null 2> $OUT.err
  python3 - $OUT $n <<'PY'
import csv, glob, sys
root, name = sys.argv[1], sys.argv[2]
dur = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
cnt = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            cnt.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            cnt[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
ids = sorted(set(dur) & set(cnt.get("GRBM_GUI_ACTIVE", {})), key=int)[-20:]   # the timed (settled) launches
if not ids:
    print(name, "no data", len(dur), {k: len(v) for k, v in cnt.items()}); sys.exit()
m = lambda k: sum(cnt[k][i] for i in ids) / len(ids)
d = sum(dur[i] for i in ids) / len(ids)
cyc = m("GRBM_GUI_ACTIVE") / 8
print("(%d launches averaged) " % len(ids), end="")
print("%-16s %.4f ms  %.3f Mcycles  clock %.3f GHz  VALU busy %.1f%%  VALU insts/wave %.0f  wait_any %.1f%% wait_inst %.1f%%" % (
    name, d / 1e6, cyc / 1e6, cyc / d, 100 * m("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc, m("SQ_INSTS_VALU") / max(m("SQ_WAVES") if "SQ_WAVES" in cnt else 98304, 1),
    100 * m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), 100 * m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES")), " MFMA busy %.1f%%" % (100 * m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc) if "SQ_VALU_MFMA_BUSY_CYCLES" in cnt else "")
PY
done
