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
# a name may be repeated: every occurrence is a process of its own (own buffers: where the driver happens to place 14 GB of
# buffers moves a launch by 2-3 %, so one process per build is one sample of that too); results are pooled per name
names = ["%s#%d" % (n, sys.argv[4:i + 4].count(n)) for i, n in enumerate(sys.argv[4:])]
res_only = os.environ.get("AB_RES_ONLY", "0")
kids = []
for n in names:
    base_n = n.split("#")[0]
    lib = os.path.join(ROOT, "gelato_amd", "libgelato_amd.so") if base_n == "main" else os.path.join(ROOT, "build", "variants", "libgel_%s.so" % base_n)
    env = dict(os.environ, GELATO_AMD_LIB=lib)
    p = subprocess.Popen([sys.executable, "-c", CHILD, wl, str(B), res_only], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "READY", n
    kids.append(p)
# a turn must be long against the power controller's time constant: with 60-ms turns a build inherited the headroom (or the
# debt) of the build before it, and the ranking depended on the ORDER of the variants (round 4: the same two builds 5 % apart
# in one run, level in the next).  AB_TURN_MS (default 1500) of launches per turn, the first third of every turn untimed, and the
# order of the variants shuffled every round.
turn_ms = float(os.environ.get("AB_TURN_MS", "1500"))
burst = max(4, int(turn_ms / max(3.4 * B / 65536, 0.05)))
ts = {n: [] for n in names}
import random
random.seed(1)
pairs = list(zip(names, kids))
for rd in range(rounds + 1):
    random.shuffle(pairs)
    for n, p in pairs:
        p.stdin.write("%d\n" % max(2, burst // 3)); p.stdin.flush()      # untimed lead-in of the turn
        p.stdout.readline()
        p.stdin.write("%d\n" % burst); p.stdin.flush()
        ms = float(p.stdout.readline())
        if rd >= 1:
            ts[n].append(ms)
for p in kids:
    p.stdin.write("0\n"); p.stdin.flush(); p.wait()
pooled = {}
for n in names:
    pooled.setdefault(n.split("#")[0], []).append(float(np.median(ts[n])))
out = {"workload": wl + (" (residual only)" if res_only == "1" else ""), "batch": B, "rounds": rounds, "launches_per_turn": burst, "turn_ms": turn_ms, "variants": {}}
for n in names:
    a = np.array(ts[n])
    # paired: each round's ratio to the first variant's turn of the same round
    ratio = a / np.array(ts[names[0]])
    out["variants"][n] = {"median_ms": float(np.median(a)), "p10_ms": float(np.percentile(a, 10)), "p90_ms": float(np.percentile(a, 90)),
                          "median_ratio_to_%s" % names[0]: float(np.median(ratio)), "evals_per_s_median": B / float(np.median(a)) * 1e3}
out["pooled"] = {k: {"processes": len(v), "median_ms_per_process": [round(x, 4) for x in v], "mean_ms": float(np.mean(v)),
                      "ratio_to_%s" % names[0].split("#")[0]: float(np.mean(v) / np.mean(pooled[names[0].split("#")[0]]))} for k, v in pooled.items()}
print(json.dumps(out))
