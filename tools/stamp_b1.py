#!/usr/bin/env python3
"""B = 1 (split latency form): where the wavefronts of ONE launch spend their lifetime (stamps of the -DGEL_STAMP build,
per-wavefront differences in shader cycles).  GPU box:
GELATO_AMD_LIB=build/variants/libgel_stamp.so python3 tools/stamp_b1.py [workload]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
from gelato_amd._lib import lib
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
pdict, unitdict, condition, xdict = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pdict, unitdict))
dev = torch.device("cuda:0")
dX = torch.from_numpy(pack_x(xdict)[None, :].copy()).to(dev)
dres = torch.empty((1, E.nres), dtype=torch.float64, device=dev)
djv = torch.empty((1, E.V), dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
L = lib()
L.gel_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    for _ in range(200):
        E.eval_batch_device(1, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(200):
        E.eval_batch_device(1, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    ev1.record(); torch.cuda.synchronize()
    n = 4096 * 8
    buf = np.zeros(n, dtype=np.uint64)
    assert L.gel_debug_stamps(buf.ctypes.data, n) == 0
    st = buf.reshape(-1, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    st = st[st[:, 7] > st[:, 0]]           # wavefronts that ran to the end (the counters of different XCDs are not synchronised: per-wavefront differences only)
    life = st[:, 7] - st[:, 0]
    print("%s B=1, %.2f us per launch (events over 200), %d wavefronts to the end; shader cycles per wavefront:" % (wl, ev0.elapsed_time(ev1) * 5.0, len(st)))
    print("  %-52s median %7.0f   max %7.0f" % ("lifetime", np.median(life), life.max()))
    for a, b, nm in ((0, 1, "entry -> descriptors read"), (1, 5, "-> D.X, mass / position / quaternion groups written"), (5, 6, "-> centre + light sweeps"), (6, 7, "-> position sweeps, end")):
        ok = (st[:, b] > 0) & (st[:, a] > 0)
        col = (st[:, b] - st[:, a])[ok]
        if len(col):
            print("  %-52s median %7.0f   max %7.0f   (%d wavefronts)" % (nm, np.median(col), col.max(), len(col)))
