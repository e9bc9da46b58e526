#!/usr/bin/env python3
"""Round trip of ONE callback through the C-ABI (gel_eval_callback: fused defect launch + row table + the three aero kinds, all
outputs to caller arrays), values only and values + derivatives, per workload.  GPU box."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import con_aero, con_dynamics, con_init_terminal_knot as ck, pack_x, problem

def t(f, n=400):
    for _ in range(30): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts = 1e6 * np.array(ts)
    return {"median_us": round(float(np.median(ts)), 1), "p10": round(float(np.percentile(ts, 10)), 1), "p90": round(float(np.percentile(ts, 90)), 1)}

for wl in sys.argv[1:] or ["example", "mixed-6x64"]:
    pdict, unitdict, condition, xdict = problem.make_problem(wl)
    E = con_dynamics.engine_of(pdict, unitdict)
    x = pack_x(xdict)
    out = {"workload": wl}
    out["defect groups only, values"] = t(lambda: E.eval_callback(x, False))
    out["defect groups only, values + derivatives"] = t(lambda: E.eval_callback(x, True))
    S = pdict["num_sections"]
    for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
        E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
    if all(k in pdict for k in ("event_index", "RocketStage")) and wl == "example":
        ck.rows_of(pdict, unitdict, condition)
    else:
        E.rows_configure([(i, 1.0, i + 1, -1.0, 0.0) for i in range(100)], [("radius", 3, 1.0, 0.0), ("speed", 9, 1.0, 0.0)])
    out["rows"] = [E._nlin, E._nfn]
    out["aero rows"] = sum(E.aero_dims(k)[0] for k in E.AERO_KINDS)
    out["defect + rows + aero, values"] = t(lambda: E.eval_callback(x, False))
    out["defect + rows + aero, values + derivatives"] = t(lambda: E.eval_callback(x, True))
    print(json.dumps(out))
