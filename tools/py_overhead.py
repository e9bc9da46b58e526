#!/usr/bin/env python3
"""(no GPU needed) Host time of the Python mirrors around ONE callback with the device call stubbed out: what objfunc / sens of
driver.make_callbacks cost besides gel_eval_callback.  usage: tools/py_overhead.py [workload] [--profile]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import con_dynamics, driver, engine, problem

wl = [a for a in sys.argv[1:] if not a.startswith("--")] or ["mixed-6x64"]
pdict, unitdict, condition, xdict = problem.make_problem(wl[0])
pdict["device"] = -1
pdict["gelato_amd_share_values"] = True
S = pdict["num_sections"]
cond = dict(condition)
if wl[0] != "example":
    cond.update({"AOA_max": {}, "dynamic_pressure_max": {}, "Q_alpha_max": {}})
real = engine.lib()


class Stub:
    def __getattr__(self, name):
        if name == "gel_eval_callback":
            return lambda *a: 0
        if name == "gel_pinned_buffers":      # a host-only handle has no pinned buffers: plain arrays stand in
            def pinned(h, r, v, x0, x1):
                self.__dict__["_keep"] = (np.zeros(E.nres), np.zeros(max(E.total_nnz, 1)), np.zeros(E.nvars), np.zeros(E.nvars))
                for q, a in zip((r, v, x0, x1), self._keep):
                    q._obj.value = a.ctypes.data
                return 0
            return pinned
        return getattr(real, name)


E = con_dynamics.engine_of(pdict, unitdict)
if wl[0] != "example":
    for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
        E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
engine.lib = lambda: Stub()
E._vals = np.zeros(E.total_nnz)
objfunc, sens = driver.make_callbacks(pdict, unitdict, cond)


def timeit(f, n=2000):
    for _ in range(50): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


funcs, _ = objfunc(xdict)
xd = [xdict, {k: v * (1 + 1e-9) for k, v in xdict.items()}]     # the optimiser hands a new point to every callback
flip = [0]


def nxt():
    flip[0] ^= 1
    return xd[flip[0]]


print(wl[0], "objfunc %.1f us, sens %.1f us (device call stubbed)" % (timeit(lambda: objfunc(nxt())), timeit(lambda: sens(nxt(), funcs))))
if "--profile" in sys.argv:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): (objfunc(nxt()) if "--obj" in sys.argv else sens(nxt(), funcs))
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
