import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gelato_amd import driver, problem, con_dynamics
pdict, unitdict, condition, xdict = problem.make_problem("mixed-6x64")
pdict["gelato_amd_share_values"] = True
objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
funcs, _ = objfunc(xdict)
xs = [{k: v * (1 + 1e-9 * i) for k, v in xdict.items()} for i in range(4)]
for i in range(50): sens(xs[i & 3], funcs)
ts = []
for i in range(400):
    t0 = time.perf_counter(); sens(xs[i & 3], funcs); ts.append(time.perf_counter() - t0)
print("sens median %.1f us  p10 %.1f" % (1e6 * np.median(ts), 1e6 * np.percentile(ts, 10)))
st = con_dynamics._state(pdict, unitdict)
E_ = st.engine
ts = []
for i in range(400):
    x = con_dynamics.pack_x(xs[i & 3], out=st._xb[i & 1])
    t0 = time.perf_counter(); E_.eval_callback(x, True, xptr=st._xp[i & 1]); ts.append(time.perf_counter() - t0)
print("eval_callback(pinned x) median %.1f us" % (1e6 * np.median(ts)))
ts = []
for i in range(400):
    t0 = time.perf_counter(); con_dynamics.pack_x(xs[i & 3], out=st._xb[i & 1]); ts.append(time.perf_counter() - t0)
print("pack_x into pinned %.1f us" % (1e6 * np.median(ts)))
pr = cProfile.Profile(); pr.enable()
for i in range(1000): sens(xs[i & 3], funcs)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
