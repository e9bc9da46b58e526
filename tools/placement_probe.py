"""(GPU box) The fused launch on the same data in buffers allocated at different places of ONE process: three distinct levels (3.06 /
3.20 / 3.30 ms at 6 x 64, B = 65536), each reproducible to 0.1 % -- what gelato_amd/placement.py chooses among.  usage: placement_probe.py [seed]"""
import os, sys, time, random, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gelato_amd import Engine, con_dynamics, pack_x, problem
pd, ud, c, xd = problem.make_problem("mixed-6x64")
E = Engine(con_dynamics.problem_arrays(pd, ud))
B = 65536
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64 + 1, 1))[:B]
s = torch.cuda.current_stream().cuda_stream
def burst(dX, r, j, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
hX = torch.from_numpy(X)
for trial in range(6):
    torch.cuda.empty_cache()
    pad = torch.empty(random.randrange(1, 4096) * (1 << 20) // 8, dtype=torch.float64, device="cuda")   # shifts what follows
    dX = hX.cuda()
    r = torch.empty((B, E.nres), dtype=torch.float64, device="cuda"); j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
    burst(dX, r, j, 100)
    ts = [burst(dX, r, j, 100) for _ in range(4)]
    print(trial, "pad %5d MB" % (pad.numel() * 8 >> 20), "ms %s" % " ".join("%.4f" % t for t in ts), "x %x res %x jv %x" % (dX.data_ptr(), r.data_ptr(), j.data_ptr()), flush=True)
    del pad, dX, r, j
