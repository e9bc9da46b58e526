#!/bin/bash
# Memory-system counters of the fused kernel (one rocprofv3 --pmc run per group; GPU box): L1 (TCP) request latencies and stalls,
# address translation, TA / SQ queue-full cycles, L2 -> fabric write stalls.  Usage: pmc_memsys.sh <tag> "<bench args>"   env: GELATO_AMD_LIB
TAG=$1; BA="$2"; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=/tmp/pmcm_$TAG; rm -rf $OUT; mkdir -p $OUT
run() { local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE "$@" --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BA > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; }; }
run a TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run b TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
run c TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
run d SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run e TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
run f TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run g TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
run h SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM
python3 - $OUT $TAG <<'PY'
import csv, glob, sys, json
root, tag = sys.argv[1], sys.argv[2]
agg = {}
for grp in "abcdefgh":
    dur = {}
    for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (root, grp), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = {}
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, grp), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" in r["Kernel_Name"]:
                d = per.setdefault(int(r["Dispatch_Id"]), {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)[-4:]
    g = {}
    for i in ids:
        for k, v in per[i].items():
            g[k] = g.get(k, 0.0) + v / len(ids)
    dd = [dur[i] for i in ids if i in dur]
    g["duration_ns"] = sum(dd) / max(len(dd), 1)
    agg[grp] = g
print("== memsys", tag)
for grp, g in agg.items():
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    print(" pass %s: %.4f ms, %.3f Mcycles" % (grp, g["duration_ns"] / 1e6, cyc / 1e6))
    for k in sorted(g):
        if k not in ("GRBM_GUI_ACTIVE", "duration_ns"):
            print("   %-40s %.5g   (per cycle %.4g, per CU-cycle %.4g)" % (k, g[k], g[k] / max(cyc, 1), g[k] / max(cyc, 1) / 256))
a = agg.get("a", {})
if a.get("TCP_TCC_READ_REQ_sum"):
    print(" mean L1->L2 read latency %.0f cycles, write latency %.0f cycles" % (a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"], a["TCP_TCC_WRITE_REQ_LATENCY_sum"] / max(a["TCP_TCC_WRITE_REQ_sum"], 1)))
gg = agg.get("g", {})
if gg.get("TCC_EA0_WRREQ_sum"):
    print(" mean L2->fabric write latency %.0f cycles, read latency %.0f cycles" % (gg["TCC_EA0_WRREQ_LEVEL_sum"] / gg["TCC_EA0_WRREQ_sum"], gg["TCC_EA0_RDREQ_LEVEL_sum"] / max(gg["TCC_EA0_RDREQ_sum"], 1)))
json.dump(agg, open("%s/gpurun_out/memsys_%s.json" % (__import__("os").environ.get("GRAFT_REPO_ROOT", "."), tag), "w"), indent=1)
PY
