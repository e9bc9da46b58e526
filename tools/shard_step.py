#!/usr/bin/env python3
"""One phase-shard step with a SIMULATED world on one GPU (GPU box): per rank the packed unit-shard kernel (HIP events), the
exchange stood in for by a device-to-device copy of the (N-1)/N of the buffer an all-gather would deliver, and the optional
one-launch gather into the ordinary layouts -- pack_us (always 0: the kernel writes its slice), kernel_us, copy_us, unpack_us,
beside the single-GPU fused launch of the same batch.  tools/shard_step.py [workload] > profiles/r04/shard_step.json"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, _lib, con_dynamics, pack_x, parallel, problem

wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
pdict, unitdict, condition, xdict = problem.make_problem(wl)
prob = con_dynamics.problem_arrays(pdict, unitdict)
E = Engine(prob)
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps      # us


out = {"workload": wl, "build": _lib.build_info(), "rows": [],
       "note": "simulated world on ONE MI355X: every rank's kernel timed alone on the whole chip (what it has on its own GPU); the "
               "all-gather is replaced by a device copy of the bytes it would deliver to one rank (a lower bound of the exchange: "
               "no xGMI latency); pack_us = 0 by construction (the kernel writes its slice of the exchange buffer), unpack_us = the "
               "optional gather into the ordinary layouts (a consumer can read through the map instead)"}
for world in (2, 4, 8):
    sh = parallel.UnitShards(E, world, 0)
    for B in (1, 64, 1024):
        X = problem.synthetic_batch(pack_x(xdict), E.M, min(B, 64))
        X = np.tile(X, (B // len(X) + 1, 1))[:B]
        dX = torch.from_numpy(X).to(dev)
        dout = sh.buffer(B, dev)
        recv = torch.empty_like(dout)
        dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
        djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
        reps = 200 if B <= 64 else 50
        k_us = [timed(lambda r=r: E.eval_shard_packed_device(B, dX.data_ptr(), dout.data_ptr(), r, s, plan=sh.plan), reps) if sh.ranges[r][1] else 0.0
                for r in range(world)]
        copy_us = timed(lambda: recv[1:].copy_(dout[1:]), reps)
        unpack_us = timed(lambda: E.shard_unpack_device(B, dout.data_ptr(), dres.data_ptr(), djv.data_ptr(), s, plan=sh.plan), reps)
        fused_us = timed(lambda: E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s), reps)
        # wall time of one rank's step as the host sees it: launch + (stand-in) exchange + synchronise
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            E.eval_shard_packed_device(B, dX.data_ptr(), dout.data_ptr(), 0, s, plan=sh.plan)
            recv[1:].copy_(dout[1:])
            torch.cuda.synchronize()
        wall_us = 1e6 * (time.perf_counter() - t0) / reps
        out["rows"].append({"world": world, "B": B, "pack_us": 0.0, "kernel_us_per_rank": [round(v, 2) for v in k_us], "kernel_us_max": round(max(k_us), 2),
                            "exchange_standin_copy_us": round(copy_us, 2), "unpack_us_optional": round(unpack_us, 2),
                            "step_us_device": round(max(k_us) + copy_us, 2), "step_wall_us_rank0_synchronised": round(wall_us, 2),
                            "single_gpu_fused_launch_us": round(fused_us, 2),
                            "bytes_received_per_rank": sh.bytes_received_per_vector() * B, "slice_doubles": sh.width,
                            "units_per_rank": [c for _, c in sh.ranges]})
print(json.dumps(out, indent=1))
