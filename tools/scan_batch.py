#!/usr/bin/env python3
"""Kernel time of the fused eval launch vs batch size / workload (HIP events on the launch stream)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem

def run(workload, Bs, reps=10):  # reps: lower bound, each measurement runs for >= ~20 ms
    pdict, unitdict, condition, xdict = problem.make_problem(workload)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    S = pdict["num_sections"]; ps = pdict["ps_params"]
    E = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)], flags=int(os.environ.get("GEL_FLAGS", "0")))
    x0 = pack_x(xdict)
    dev = torch.device("cuda:0")
    Bmax = max(Bs)
    X = problem.synthetic_batch(x0, E.M, min(Bmax, 256))
    X = np.tile(X, (Bmax // len(X) + 1, 1))[:Bmax]
    dX = torch.from_numpy(X).to(dev)
    dres = torch.empty((Bmax, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((Bmax, E.V), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    out = []
    # settle the power state first: from idle the chip boosts, overshoots into a ~10 ms dip, then settles
    t_end = __import__("time").time() + float(os.environ.get("SCAN_SETTLE_S", "0.15"))
    while __import__("time").time() < t_end:
        E.eval_batch_device(min(Bmax, 4096), dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
        torch.cuda.synchronize()
    for B in Bs:
        for jac in (True, False):
            reps = max(10, int(20.0 / max(0.4 * B / 4096, 1e-3)))       # >= ~20 ms per measurement
            for _ in range(3):
                E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr() if jac else 0, s)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr() if jac else 0, s)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            out.append({"workload": workload, "B": B, "jac": jac, "ms": ms, "evals_per_s": B / ms * 1e3,
                        "GBps": (E.algorithmic_bytes if jac else 8 * (E.nvars + E.nres)) * B / ms / 1e6})
            print(json.dumps(out[-1]), flush=True)
    return out

if __name__ == "__main__":
    p = torch.cuda.get_device_properties(0)
    print(p.name, "CUs", p.multi_processor_count, "mem GB", p.total_memory / 2**30, flush=True)
    wl = sys.argv[1:] or ["mixed-6x64", "dense-6x64"]
    for w in wl:
        run(w, [int(b) for b in os.environ.get("SCAN_B", "1,8,64,128,256,512,1024,2048,4096,8192").split(",")])
