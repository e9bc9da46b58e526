#!/usr/bin/env python3
"""fp64 work of the fused kernel per launch, from two rocprofv3 --pmc passes of tools/gpu_record.sh (instruction counters; cycles):
   flops = 64 * (ADD_F64 + MUL_F64 + 2 FMA_F64 [+ TRANS_F64]) + 2048 * MFMA_F64 instructions (v_mfma_f64_16x16x4_f64: 16 x 16 x 4 FMAs)
and the time the fp64 datapath was occupied (vector fp64 and matrix fp64 share it: SQ_ACTIVE_INST_VALU * 4 + SQ_VALU_MFMA_BUSY_CYCLES
per SIMD, against GRBM_GUI_ACTIVE / 8 cycles).  Usage: pmc_fp64.py <dir> [bench args]"""
import csv, glob, json, os, sys


def launch_policy(workload, batch, resonly):
    """what the HOST side decided for this launch (instantiation, wavefronts, LDS order, store policy: gel_launch_info + the
    environment switches that change it) -- recorded with the counters, so that bench.py does not report them for a library whose
    device code is the same but whose launch policy is not"""
    from gelato_amd import Engine, con_dynamics, problem
    pd, ud, _c, _x = problem.make_problem(workload)
    E = Engine(con_dynamics.problem_arrays(pd, ud))
    return {"launch_info": E.launch_info(batch, True, not resonly), "num_chunks": E.num_chunks()}



root = sys.argv[1]
args = sys.argv[2:]


def opt(name, default):
    return args[args.index(name) + 1] if name in args else default


def means(sub):
    per, name, dur = {}, None, {}
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" not in r.get("Kernel_Name", ""):
                continue
            name = r["Kernel_Name"]
            per.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per, key=int)[-4:]          # the timed launches
    out = {}
    for i in ids:
        for k, v in per[i].items():
            out[k] = out.get(k, 0.0) + v / len(ids)
    d = [dur[i] for i in ids if i in dur]
    return out, name, (sum(d) / len(d) if d else None), len(ids)


a, kname, dur_a, na = means("pmc_fp64a")
b, _, dur_b, nb = means("pmc_fp64b")
out = {"workload": opt("--workload", "mixed-6x64") + ("_resonly" if "--residual-only" in args else ""), "batch": int(opt("--batch", "65536")),
       "kernel": kname, "launches_averaged": [na, nb]}
if a and b:
    valu_flops = 64.0 * (a.get("SQ_INSTS_VALU_ADD_F64", 0) + a.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * a.get("SQ_INSTS_VALU_FMA_F64", 0)
                         + a.get("SQ_INSTS_VALU_TRANS_F64", 0))
    mfma_flops = 2048.0 * a.get("SQ_INSTS_VALU_MFMA_F64", a.get("SQ_INSTS_MFMA", 0))
    cyc = b["GRBM_GUI_ACTIVE"] / 8.0
    out.update({
        "fp64_instructions_per_launch": {k: a.get(k) for k in sorted(a)},
        "fp64_flops_per_launch": valu_flops + mfma_flops, "valu_fp64_flops": valu_flops, "mfma_fp64_flops": mfma_flops,
        "cycles_per_launch": cyc, "kernel_ns_profiled": dur_b, "clock_ghz": cyc / dur_b if dur_b else None,
        "waves": b.get("SQ_WAVES"), "valu_instructions_per_wave": b["SQ_INSTS_VALU"] / b["SQ_WAVES"],
        "valu_busy": b["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, "mfma_busy": b["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
        "wait_inst_share": b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"], "wait_any_share": b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"]})
    out["fp64_pipe_busy"] = out["valu_busy"] + out["mfma_busy"]
out["source"] = ("tools/gpu_record.sh -> rocprofv3 --kernel-trace --pmc, two passes (fp64 instruction counters; cycles and busy counters), "
                 "bench.py --steps 4 --warmup 1 --settle-ms 0; profiled passes run at a lower clock than un-profiled ones")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gelato_amd import _lib  # noqa: E402  (provenance only: which build these counters describe)
out.update({"build_" + k: v for k, v in _lib.build_info().items()})
try:
    out["launch_policy"] = launch_policy(out["workload"].replace("_resonly", ""), out["batch"], out["workload"].endswith("_resonly"))
except Exception as ex:  # noqa: BLE001
    out["launch_policy"] = {"error": str(ex)[:200]}
print(json.dumps(out, indent=1))
