#!/usr/bin/env python3
"""Does the launch time depend on WHERE the buffers lie?  One process, the fused launch at B = 65536, re-allocating x / res / jvar
several times with different paddings in front of them (torch's caching allocator emptied in between), 2 s of launches per
placement, twice round the set.  GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64 + 1, 1))[:B]
s = torch.cuda.current_stream().cuda_stream
pads = [0, 1 << 20, 3 << 20, (1 << 30) + (5 << 20), 7 << 12, (2 << 30) + 12345 * 256]
rows = []
for rd in range(2):
    for pad in pads:
        torch.cuda.empty_cache()
        p0 = torch.empty(max(pad, 1), dtype=torch.uint8, device="cuda")
        dX = torch.from_numpy(X).cuda()
        p1 = torch.empty(max(pad // 3, 1), dtype=torch.uint8, device="cuda")
        r = torch.empty((B, E.nres), dtype=torch.float64, device="cuda")
        p2 = torch.empty(max(pad // 7, 1), dtype=torch.uint8, device="cuda")
        j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
        for _ in range(100): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 50)
        rows.append({"round": rd, "pad": pad, "x": hex(dX.data_ptr()), "res": hex(r.data_ptr()), "jvar": hex(j.data_ptr()), "median_ms": float(np.median(ts)), "min_ms": min(ts), "max_ms": max(ts)})
        print(json.dumps(rows[-1]), flush=True)
        del dX, r, j, p0, p1, p2
