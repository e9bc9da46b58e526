#!/usr/bin/env python3
"""The drop-in figure: wall time of objfunc / sens through the reference-named Python functions
(gelato_amd.con_dynamics + driver.make_callbacks), per call, on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import driver, problem
for name in sys.argv[1:] or ["example", "mixed-6x64", "dense-6x64"]:
    pdict, unitdict, condition, xdict = problem.make_problem(name)
    objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
    driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=3)
    st = driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=50)
    pdict["gelato_amd_share_values"] = True
    st2 = driver.mock_optimizer_loop(objfunc, sens, xdict, iterations=50)
    # medians per call beside the reference's totals (driver.mock_optimizer_loop sums like Trajectory_Optimization.py:511-517)
    import time
    import numpy as np
    x = {k: v.copy() for k, v in xdict.items()}
    to, ts = [], []
    for it in range(200):
        for k in x:
            x[k] = x[k] * (1.0 + 1e-7)
        a = time.perf_counter(); funcs, _f = objfunc(x); b = time.perf_counter(); sens(x, funcs); c = time.perf_counter()
        to.append(b - a); ts.append(c - b)
    print(json.dumps({"workload": name, "sens_median_ms_shared_values": round(1e3 * float(np.median(ts)), 4), "objfunc_median_ms_shared_values": round(1e3 * float(np.median(to)), 4), "userSensTime_ms_per_call_shared_values": round(1e3 * st2["userSensTime"] / st2["userSensCalls"], 4), "userObjTime_ms_per_call": round(1e3 * st["userObjTime"] / st["userObjCalls"], 4),
                      "userSensTime_ms_per_call": round(1e3 * st["userSensTime"] / st["userSensCalls"], 4), "fails": st["fails"]}))
