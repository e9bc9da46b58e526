import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gelato_amd import Engine, con_dynamics, pack_x, problem
for wl in ("stress-12x128", "mixed-6x64"):
    pdict, unitdict, _, xdict = problem.make_problem(wl)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    E = Engine(prob)
    x0 = pack_x(xdict)
    for _ in range(3): E.jac_fd("vel", x0 * (1 + 1e-9 * np.random.rand()))
    ts = []
    for i in range(8):
        x = x0 * (1 + 1e-9 * (i + 1))
        t0 = time.perf_counter(); E.jac_fd("vel", x); ts.append(time.perf_counter() - t0)
    print(wl, "jac_fd vel ms: median %.2f min %.2f" % (1e3 * np.median(ts), 1e3 * min(ts)))
