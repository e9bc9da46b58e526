import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gelato_amd import driver, problem
pdict, unitdict, condition, xdict = problem.make_problem("mixed-6x64")
objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
x = {k: v.copy() for k, v in xdict.items()}
ts = []
for it in range(60):
    for k in x: x[k] = x[k] * (1.0 + 1e-7)
    f, _ = objfunc(x)
    t0 = time.perf_counter(); fs, _ = sens(x, f); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.2f" % t for t in ts))
