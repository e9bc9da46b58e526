"""(GPU box) SURVEY 8(d)'s batch scan: the fused launch at B in {1, 64, 1024, 8192, 65536}, inputs resident in HBM, with and without the
D2H of the results (residual rows + compact Jacobian values) on the same stream; >= 1000 evals per row after warm-up, HIP events.
B = 1 .. 64 run the split latency form and are launch-latency bound.  usage: batch_scan.py [workload]  -> one JSON document"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gelato_amd import Engine, _lib, con_dynamics, pack_x, problem
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
x0 = pack_x(xd)
rows = []
for B in (1, 64, 1024, 8192, 65536):
    X = np.tile(problem.synthetic_batch(x0, E.M, min(B, 256)), (B // 256 + 1, 1))[:B]
    dX = torch.from_numpy(X).to(dev)
    r = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    j = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    hr = torch.empty((B, E.nres), dtype=torch.float64).pin_memory()
    hj = torch.empty((B, E.V), dtype=torch.float64).pin_memory()
    info = E.launch_info(B, True, True)
    form = "split (latency form)" if info[2] else ("cooperative, D.X on the matrix pipe" + (", two vectors per wavefront" if info[4] else ""))
    for d2h in (False, True):
        def step():
            E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
            if d2h:
                hr.copy_(r, non_blocking=True); hj.copy_(j, non_blocking=True)
        n = max(8, int(np.ceil(1000 / B)))
        if B >= 8192 and d2h:
            n = 4
        for _ in range(max(3, min(200, n // 4))):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rows.append({"B": B, "d2h_of_results": d2h, "kernel_form": form, "steps_timed": n, "evals_timed": n * B, "ms_per_step": ms,
                     "us_per_eval": 1e3 * ms / B, "evals_per_s": B / (ms * 1e-3),
                     "hbm_frac_algorithmic": E.algorithmic_bytes * B / (ms * 1e-3) / 8e12,
                     "d2h_bytes_per_step": (E.stored_bytes * B) if d2h else 0})
    del dX, r, j, hr, hj
print(json.dumps({"workload": wl, "build": _lib.build_info(), "algorithmic_bytes_per_eval": E.algorithmic_bytes,
                  "stored_bytes_per_eval": E.stored_bytes,
                  "note": "B = 1 .. 64 run the split latency form and are launch-latency bound (SURVEY 8d); the D2H rows move the residual rows + "
                          "compact values over PCIe on the same stream (pinned destination)", "rows": rows}))
