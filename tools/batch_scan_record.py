#!/usr/bin/env python3
"""the fused launch over the batch size, B in {1, 64, 1024, 8192, 65536}, with and without the D2H copy of the
variable Jacobian entries and residuals (HIP events on the launch stream, settled power state, >= 1000 evals per point).
GPU box:  python3 tools/batch_scan_record.py [workload] > profiles/r04/batch_scan.json"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, _lib, con_dynamics, pack_x, problem

wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
pdict, unitdict, condition, xdict = problem.make_problem(wl)
prob = con_dynamics.problem_arrays(pdict, unitdict)
E = Engine(prob)
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
Bs = [1, 64, 1024, 8192, 65536]
Bmax = max(Bs)
X = problem.synthetic_batch(pack_x(xdict), E.M, 256)
X = np.tile(X, (Bmax // len(X) + 1, 1))[:Bmax]
dX = torch.from_numpy(X).to(dev)
dres = torch.empty((Bmax, E.nres), dtype=torch.float64, device=dev)
djv = torch.empty((Bmax, E.V), dtype=torch.float64, device=dev)
hres = torch.empty((Bmax, E.nres), dtype=torch.float64).pin_memory()
hjv = torch.empty((Bmax, E.V), dtype=torch.float64).pin_memory()
t_end = time.time() + 0.3
while time.time() < t_end:      # settle the power state
    E.eval_batch_device(4096, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    torch.cuda.synchronize()
rows = []
for B in Bs:
    info = E.launch_info(B, True, True)
    form = "split (latency form)" if info[2] else "cooperative, D.X on the matrix pipe"
    for d2h in (False, True):
        reps = max(10, min(2000, int(3000 / B) + 1, int(2e9 / (B * 8 * (E.V + E.nres))) + 1 if d2h else 10**9))

        def step():
            E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
            if d2h:
                hres[:B].copy_(dres[:B], non_blocking=True)
                hjv[:B].copy_(djv[:B], non_blocking=True)
        for _ in range(3):
            step()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            step()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        rows.append({"B": B, "d2h_of_results": d2h, "kernel_form": form, "steps_timed": reps, "evals_timed": reps * B, "ms_per_step": ms,
                     "us_per_eval": 1e3 * ms / B, "evals_per_s": B / ms * 1e3, "hbm_frac_algorithmic": E.algorithmic_bytes * B / (ms * 1e-3) / 8e12,
                     "d2h_bytes_per_step": 8 * (E.V + E.nres) * B if d2h else 0})
print(json.dumps({"workload": wl, "build": _lib.build_info(), "algorithmic_bytes_per_eval": E.algorithmic_bytes, "stored_bytes_per_eval": E.stored_bytes,
                  "note": "B = 1 .. 64 run the split latency form and are launch-latency bound (SURVEY 8d says so in every report); "
                          "the D2H rows move the compact values + residuals over PCIe on the same stream",
                  "rows": rows}, indent=1))
