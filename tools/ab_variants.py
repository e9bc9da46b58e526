#!/usr/bin/env python3
"""A/B of engine builds on ONE box, drift-free: one child process per library (its own buffers, resident for the whole run), the
parent lets them take turns -- one short burst of launches each, round robin, many rounds -- so that the thermal / power state
of the chip (which moves a launch by several per cent within seconds) hits every variant alike; medians per variant.
GPU box:  python3 tools/ab_variants.py <workload> <batch> <rounds> main nox ...   [env AB_RES_ONLY=1]   (main = the in-tree library)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
wl, B, res_only = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64 + 1, 1))[:B]
dX = torch.from_numpy(X).cuda()
r = torch.empty((B, E.nres), dtype=torch.float64, device="cuda"); j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
jp = 0 if res_only else j.data_ptr()
for _ in range(5): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), jp, s)
torch.cuda.synchronize()
print("READY", flush=True)
for line in sys.stdin:
    n = int(line)
    if n <= 0: break
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), jp, s)
    b.record(); torch.cuda.synchronize()
    print(a.elapsed_time(b) / n, flush=True)
""" % ROOT
wl, B, rounds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
names = sys.argv[4:]
res_only = os.environ.get("AB_RES_ONLY", "0")
kids = []
for n in names:
    lib = os.path.join(ROOT, "gelato_amd", "libgelato_amd.so") if n == "main" else os.path.join(ROOT, "build", "variants", "libgel_%s.so" % n)
    env = dict(os.environ, GELATO_AMD_LIB=lib)
    p = subprocess.Popen([sys.executable, "-c", CHILD, wl, str(B), res_only], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "READY", n
    kids.append(p)
burst = max(4, int(60.0 / max(3.4 * B / 65536, 0.05)))      # ~60 ms of launches per turn
ts = {n: [] for n in names}
for rd in range(rounds + 2):
    for n, p in zip(names, kids):
        p.stdin.write("%d\n" % burst); p.stdin.flush()
        ms = float(p.stdout.readline())
        if rd >= 2:
            ts[n].append(ms)
for p in kids:
    p.stdin.write("0\n"); p.stdin.flush(); p.wait()
base = np.median(ts[names[0]])
out = {"workload": wl + (" (residual only)" if res_only == "1" else ""), "batch": B, "rounds": rounds, "launches_per_turn": burst, "variants": {}}
for n in names:
    a = np.array(ts[n])
    # paired: each round's ratio to the first variant's turn of the same round
    ratio = a / np.array(ts[names[0]])
    out["variants"][n] = {"median_ms": float(np.median(a)), "p10_ms": float(np.percentile(a, 10)), "p90_ms": float(np.percentile(a, 90)),
                          "median_ratio_to_%s" % names[0]: float(np.median(ratio)), "evals_per_s_median": B / float(np.median(a)) * 1e3}
print(json.dumps(out))
