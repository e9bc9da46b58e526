#!/usr/bin/env python3
"""Where the fused launch spends its time by phase type: the work items of every run of consecutive phases of one type
(aerodynamic / NoAir) launched alone (gel_eval_shard_device), against the whole launch.  GPU box."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem

def main(workload, B):
    pdict, unitdict, condition, xdict = problem.make_problem(workload)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    E = Engine(prob)
    x0 = pack_x(xdict)
    dev = torch.device("cuda:0")
    X = problem.synthetic_batch(x0, E.M, min(B, 256))
    X = np.tile(X, (B // len(X) + 1, 1))[:B]
    dX = torch.from_numpy(X).to(dev)
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    cph = E.chunk_phase()
    air = (np.asarray(prob["reference_area"]) != 0.0)[cph]
    runs, a = [], 0
    for i in range(1, len(cph) + 1):
        if i == len(cph) or air[i] != air[a]:
            runs.append((a, i - a, bool(air[a])))
            a = i
    def timeit(f, reps=40):
        for _ in range(5): f()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps): f()
        t1.record(); torch.cuda.synchronize()
        return t0.elapsed_time(t1) / reps
    for _ in range(200):
        E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    torch.cuda.synchronize()
    full = timeit(lambda: E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s))
    parts = {"air": 0.0, "noair": 0.0}
    items = {"air": 0, "noair": 0}
    for (c0, cnt, is_air) in runs:
        ms = timeit(lambda: E.eval_shard_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), c0, cnt, s))
        parts["air" if is_air else "noair"] += ms
        items["air" if is_air else "noair"] += cnt
    print(json.dumps({"workload": workload, "B": B, "full_ms": full, "air_ms": parts["air"], "noair_ms": parts["noair"],
                      "items": items, "runs": runs}))

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
