#!/usr/bin/env python3
"""Shader clock and package power under the aero path-constraint kernel (B = 16384), like tools/power_clock.py.  GPU box."""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.power_clock import sample  # noqa: E402
code = ("import sys, time; sys.path.insert(0, %r); import numpy as np, torch\n"
        "from gelato_amd import Engine, con_dynamics, pack_x, problem\n"
        "pd, ud, c, xd = problem.make_problem('mixed-6x64'); E = Engine(con_dynamics.problem_arrays(pd, ud)); B = 16384; S = len(E.num_nodes)\n"
        "for kind, lim in (('alpha', 0.2), ('q', 4.0e4), ('qalpha', 5.0e3)): E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])\n"
        "dims = [E.aero_dims(k) for k in E.AERO_KINDS]\n"
        "X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64, 1)); dX = torch.from_numpy(X).cuda()\n"
        "dcon = [torch.empty((B, d[0]), dtype=torch.float64, device='cuda') for d in dims]; djac = [torch.empty((B, sum(d[1])), dtype=torch.float64, device='cuda') for d in dims]\n"
        "cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]; s = torch.cuda.current_stream().cuda_stream\n"
        "print('READY', flush=True); t0 = time.time(); n = 0\n"
        "while time.time() - t0 < 8.0:\n"
        "    for _ in range(50): E.eval_aero_all_device(B, dX.data_ptr(), cp, jp, s)\n"
        "    torch.cuda.synchronize(); n += 50\n"
        "print('VECTORS_PER_S', n * B / (time.time() - t0), flush=True)\n") % ROOT
pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
assert pr.stdout.readline().strip() == "READY"
time.sleep(3.0)
samples = [sample() for _ in range(4) if not time.sleep(0.8)]
rate = None
for line in pr.stdout:
    if line.startswith("VECTORS_PER_S"):
        rate = float(line.split()[1])
pr.wait()
print(json.dumps({"kernel": "aero_kernel, three kinds, 975 rows, B = 16384", "vectors_per_s": rate, "samples_under_load": samples}, indent=1))
