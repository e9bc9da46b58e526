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
    print(json.dumps({"workload": name, "userSensTime_ms_per_call_shared_values": round(1e3 * st2["userSensTime"] / st2["userSensCalls"], 4), "userObjTime_ms_per_call": round(1e3 * st["userObjTime"] / st["userObjCalls"], 4),
                      "userSensTime_ms_per_call": round(1e3 * st["userSensTime"] / st["userSensCalls"], 4), "fails": st["fails"]}))
