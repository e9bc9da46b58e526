#!/usr/bin/env python3
"""Durations of N consecutive launches of the fused kernel starting from an idle GPU (HIP events around
every launch): shows the boost -> sustained clock transition.  GPU box only."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem

workload = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
B = int(os.environ.get("SERIES_B", "4096")); n = int(os.environ.get("SERIES_N", "24"))
pdict, unitdict, condition, xdict = problem.make_problem(workload)
prob = con_dynamics.problem_arrays(pdict, unitdict)
S = pdict["num_sections"]; ps = pdict["ps_params"]
E = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)])
dev = torch.device("cuda:0")
X = problem.synthetic_batch(pack_x(xdict), E.M, 256)
dX = torch.from_numpy(np.tile(X, (B // 256 + 1, 1))[:B]).to(dev)
dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
torch.cuda.synchronize()
for trial in range(2):
    time.sleep(0.5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
        ev[i + 1].record()
    torch.cuda.synchronize()
    print(json.dumps({"workload": workload, "B": B, "us": [round(1e3 * ev[i].elapsed_time(ev[i + 1]), 1) for i in range(n)]}), flush=True)
