#!/usr/bin/env python3
"""Where the time of one B = 1 callback goes (GPU box): full gel_eval, batch(1) without the scatter,
residual only, resident launch + sync, and the kernel alone."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem

def bench(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return {"median_us": round(float(np.median(ts)), 1), "p10": round(float(np.percentile(ts, 10)), 1), "p90": round(float(np.percentile(ts, 90)), 1)}

for workload in sys.argv[1:] or ["example", "mixed-6x64"]:
    pdict, unitdict, condition, xdict = problem.make_problem(workload)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    S = pdict["num_sections"]; ps = pdict["ps_params"]
    E = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)])
    x0 = pack_x(xdict)
    _, vals, _ = E.eval(x0)
    out = {"workload": workload}
    out["gel_eval (res + full COO values)"] = bench(lambda: E.eval(x0, out=vals))
    pres, pvals = E.pinned_buffers()
    out["gel_eval into the handle's pinned buffers (zero-copy)"] = bench(lambda: E.eval(x0, out=pvals, res_out=pres))
    assert np.array_equal(pvals, vals) and np.array_equal(pres, E.eval(x0, out=vals)[0])
    out["gel_eval_callback (defect groups, values + derivatives)"] = bench(lambda: E.eval_callback(x0, True))
    out["gel_eval_callback (defect groups, values)"] = bench(lambda: E.eval_callback(x0, False))
    out["gel_eval_batch B=1 (res + compact)"] = bench(lambda: E.eval_batch(x0))
    out["gel_eval_residual"] = bench(lambda: E.eval_residual(x0))
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(x0[None, :]).to(dev)
    dres = torch.empty((1, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((1, E.V), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def resident():
        E.eval_batch_device(1, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
        torch.cuda.synchronize()
    out["resident launch + sync"] = bench(resident)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        E.eval_batch_device(1, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    b.record(); torch.cuda.synchronize()
    out["kernel alone (events, back to back)"] = round(1e3 * a.elapsed_time(b) / 200, 1)
    print(json.dumps(out, indent=1), flush=True)
