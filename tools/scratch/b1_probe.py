import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
pd, ud, c, xd = problem.make_problem("mixed-6x64")
E = Engine(con_dynamics.problem_arrays(pd, ud))
x0 = pack_x(xd)
def t(fn, n=200):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return 1e6 * (time.perf_counter() - t0) / n
pres, pvals = E.pinned_buffers()
zc = lambda: E.eval(x0, out=pvals, res_out=pres)
vals = E.eval(x0)[1]
ca = lambda: E.eval(x0, out=vals)
print("fresh process: zero-copy %.1f us, caller arrays %.1f us" % (t(zc), t(ca)))
big = torch.empty((65536, 5065 + 4224 + 16966), dtype=torch.float64, device="cuda")
print("13.8 GB resident: zero-copy %.1f us, caller arrays %.1f us" % (t(zc), t(ca)))
B = 65536
X = np.tile(problem.synthetic_batch(x0, E.M, 64), (B // 64, 1))
dX = torch.from_numpy(X).cuda(); dr = torch.empty((B, E.nres), dtype=torch.float64, device="cuda"); dj = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(30): E.eval_batch_device(B, dX.data_ptr(), dr.data_ptr(), dj.data_ptr(), s)
torch.cuda.synchronize()
print("after 30 big launches: zero-copy %.1f us, caller arrays %.1f us" % (t(zc), t(ca)))
time.sleep(1.0)
print("1 s later: zero-copy %.1f us, caller arrays %.1f us" % (t(zc), t(ca)))
E2 = Engine(con_dynamics.problem_arrays(pd, ud))
p2r, p2v = E2.pinned_buffers()
print("a second engine now: zero-copy %.1f us" % t(lambda: E2.eval(x0, out=p2v, res_out=p2r)))
print("steady state, chunks of 1000 zero-copy calls:", [round(t(zc, 1000), 1) for _ in range(8)])
time.sleep(2.0)
print("after 2 s idle:", [round(t(zc, 250), 1) for _ in range(8)])
cb = lambda: E.eval_callback(x0, True)
print("eval_callback jac:", [round(t(cb, 500), 1) for _ in range(4)])
