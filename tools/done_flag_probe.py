"""(GPU box) When do the results of a one-vector launch reach pinned host memory, and when does hipStreamSynchronize return?  The host
polls 36 sample entries of the result buffers after an asynchronous launch, then synchronises: results 25 us after the call, the
runtime's wait back after 29 us -- the 4 us the self-signalling launches (gel_eval_kernel.h signal_done) do not wait for."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import Engine, con_dynamics, pack_x, problem
pd, ud, c, xd = problem.make_problem("mixed-6x64")
E = Engine(con_dynamics.problem_arrays(pd, ud))
x = torch.from_numpy(pack_x(xd)).pin_memory()
res = torch.full((E.nres,), float("nan"), dtype=torch.float64).pin_memory()
jv = torch.full((E.V,), float("nan"), dtype=torch.float64).pin_memory()
s = torch.cuda.current_stream().cuda_stream
rn, jn = res.numpy(), jv.numpy()
for _ in range(20):
    E.eval_batch_device(1, x.data_ptr(), res.data_ptr(), jv.data_ptr(), s); E.sync(s)
ridx = np.linspace(0, E.nres - 1, 12).astype(int); jidx = np.linspace(0, E.V - 1, 24).astype(int)
lates = []
out = []
for it in range(300):
    rn.fill(np.nan); jn.fill(np.nan)
    t0 = time.perf_counter()
    E.eval_batch_device(1, x.data_ptr(), res.data_ptr(), jv.data_ptr(), s)
    t1 = time.perf_counter()
    # poll until no NaN is left anywhere (cheap checks first: the ends, then everything)
    while True:
        if not (np.isnan(rn[ridx]).any() or np.isnan(jn[jidx]).any()):
            break
    t2 = time.perf_counter()
    late = int(np.isnan(rn).sum() + np.isnan(jn).sum())
    lates.append(late)
    while np.isnan(rn).any() or np.isnan(jn).any():
        pass
    E.sync(s)
    t3 = time.perf_counter()
    out.append((t1 - t0, t2 - t0, t3 - t0))
a = np.array(out) * 1e6
print("entries still missing when the 36 samples had arrived: median %d max %d" % (np.median(lates), max(lates)))
print("launch returns %.1f us, all data visible on the host %.1f us, sync returns %.1f us (medians); sync - data p10/p50/p90 %s" % (
    np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), np.round(np.percentile(a[:, 2] - a[:, 1], [10, 50, 90]), 1)))
# reference: launch + sync without polling
out = []
for it in range(300):
    t0 = time.perf_counter()
    E.eval_batch_device(1, x.data_ptr(), res.data_ptr(), jv.data_ptr(), s); E.sync(s)
    out.append(time.perf_counter() - t0)
print("launch + sync alone %.1f us" % (1e6 * np.median(out)))
