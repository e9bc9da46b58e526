#!/usr/bin/env python3
"""B = 1 latency of the aero path constraints (three kinds, values + gradients, one launch) and of the row table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import Engine, con_dynamics, pack_x, problem
pdict, unitdict, condition, xdict = problem.make_problem(sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64")
prob = con_dynamics.problem_arrays(pdict, unitdict)
E = Engine(prob)
x0 = pack_x(xdict)
S = len(prob["num_nodes"])
for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
    E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
def t(f, n=300):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return 1e6 * (time.perf_counter() - t0) / n
print("rows", sum(E.aero_dims(k)[0] for k in E.AERO_KINDS))
print("three kinds, values + gradients : %.1f us" % t(lambda: E.eval_aero_all(x0)))
print("three kinds, values only        : %.1f us" % t(lambda: E.eval_aero_all(x0, want_jac=False)))
print("three kinds, + gradients, reuse : %.1f us" % t(lambda: E.eval_aero_all(x0, reuse=True)))
print("one kind (alpha), + gradients   : %.1f us" % t(lambda: E.eval_aero("alpha", x0)))
print("defect residual + full COO      : %.1f us" % t(lambda: E.eval(x0)))
print("defect residual only            : %.1f us" % t(lambda: E.eval_residual(x0)))
import torch
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
dX = torch.from_numpy(x0.reshape(1, -1)).to(dev)
dims = [E.aero_dims(k) for k in E.AERO_KINDS]
dcon = [torch.empty((1, d[0]), dtype=torch.float64, device=dev) for d in dims]
djac = [torch.empty((1, sum(d[1])), dtype=torch.float64, device=dev) for d in dims]
def ev(f, n=200):
    for _ in range(10): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n
cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]
print("device B=1 kernel, values only   : %.1f us" % ev(lambda: E.eval_aero_all_device(1, dX.data_ptr(), cp, None, s)))
print("device B=1 kernel, + gradients   : %.1f us" % ev(lambda: E.eval_aero_all_device(1, dX.data_ptr(), cp, jp, s)))
print("device B=1 kernel, alpha only +g : %.1f us" % ev(lambda: E.eval_aero_all_device(1, dX.data_ptr(), [cp[0], 0, 0], [jp[0], 0, 0], s)))
import ctypes as C
from gelato_amd._lib import lib
L = lib(); _dp = C.POINTER(C.c_double)
con = [np.empty((1, d[0])) for d in dims]; jac = [np.empty((1, sum(d[1]))) for d in dims]
def call(mask_c, mask_j):
    cp_ = (_dp * 3)(*[con[i].ctypes.data_as(_dp) if mask_c[i] else None for i in range(3)])
    jp_ = (_dp * 3)(*[jac[i].ctypes.data_as(_dp) if mask_j[i] else None for i in range(3)])
    xx = x0.ctypes.data_as(_dp)
    return lambda: L.gel_eval_aero_all(E._h, 1, xx, cp_, jp_)
for mc, mj in [((1,1,1),(0,0,0)), ((1,0,0),(1,0,0)), ((0,1,0),(0,1,0)), ((0,0,1),(0,0,1)), ((1,1,0),(1,1,0)), ((1,0,1),(1,0,1)), ((1,1,1),(1,1,1))]:
    print("raw C-ABI call con", mc, "jac", mj, ": %.1f us" % t(call(mc, mj)))
