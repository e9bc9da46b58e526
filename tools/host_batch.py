#!/usr/bin/env python3
"""PCIe-inclusive throughput of gel_eval_batch (pageable host arrays in and out) vs batch size (GPU box)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import Engine, con_dynamics, pack_x, problem
workload = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
pdict, unitdict, condition, xdict = problem.make_problem(workload)
prob = con_dynamics.problem_arrays(pdict, unitdict)
S = pdict["num_sections"]; ps = pdict["ps_params"]
E = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)])
X0 = problem.synthetic_batch(pack_x(xdict), E.M, 64)
for B in [int(b) for b in os.environ.get("HB", "64,128,512,2048").split(",")]:
    X = np.tile(X0, (B // 64 + 1, 1))[:B]
    r, j, _ = E.eval_batch(X)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); E.eval_batch(X, out=(r, j)); ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    print(json.dumps({"workload": workload, "B": B, "ms": round(1e3 * dt, 2), "evals_per_s": round(B / dt),
                      "GBps_moved": round(B * E.algorithmic_bytes / dt / 1e9, 2)}), flush=True)
