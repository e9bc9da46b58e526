#!/bin/bash
# per-KERNEL counter means of one bench command (several fused-kernel instantiations per step since the NoAir work
# items have a launch of their own).  Usage: pmc_kernels.sh "<bench args>"      (GPU box; one rocprofv3 --pmc run per group)
R=$GRAFT_REPO_ROOT; BA="$1"
export TMPDIR=/tmp
cd /tmp
OUT=/tmp/pmck; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $BA > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_BUSY_CYCLES
run c TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
python3 - $OUT <<'PY'
import csv, glob, sys, re
root = sys.argv[1]
agg = {}
for grp in "abc":
    dur = {}
    for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (root, grp), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, grp), recursive=True):
        rows = list(csv.DictReader(open(f)))
        ids = {}
        for r in rows:
            if "eval_kernel" in r["Kernel_Name"]:
                ids.setdefault(r["Kernel_Name"], set()).add(int(r["Dispatch_Id"]))
        keep = {k: set(sorted(v)[-10:]) for k, v in ids.items()}     # the timed (settled) launches
        for r in rows:
            k = r["Kernel_Name"]
            if k in keep and int(r["Dispatch_Id"]) in keep[k]:
                m = re.search(r"eval_kernel<([^>]*)>", k)
                a = agg.setdefault(m.group(1) if m else k, {})
                a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / len(keep[k])   # mean over the launches kept
        if grp == "a":
            for k, v in keep.items():
                m = re.search(r"eval_kernel<([^>]*)>", k)
                d = [dur[str(i)][1] for i in v if str(i) in dur]
                agg.setdefault(m.group(1) if m else k, {})["duration_ns"] = sum(d) / max(len(d), 1)
for k, a in agg.items():
    print("== eval_kernel<%s>" % k)
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc:
        print("  %.4f ms  %.3f Mcycles  clock %.2f GHz  waves %.0f  VALU busy %.1f%%  MFMA busy %.1f%%  VALU/wave %.0f  wait_inst %.1f%% wait_any %.1f%%" % (
            a["duration_ns"] / 1e6, cyc / 1e6, cyc / a["duration_ns"], a["SQ_WAVES"], 100 * a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
            100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, a["SQ_INSTS_VALU"] / a["SQ_WAVES"], 100 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]))
    for c in sorted(a):
        print("  %-32s %.5g" % (c, a[c]))
    if a.get("SQ_INSTS_VMEM_WR"):
        n = a["SQ_INSTS_VMEM_WR"] + a["SQ_INSTS_VMEM_RD"]
        print("  mean VMEM latency (level/insts) %.0f cycles; LDS %.0f" % (a["SQ_INST_LEVEL_VMEM"] / n, a["SQ_INST_LEVEL_LDS"] / max(a["SQ_INSTS_LDS"], 1)))
PY
