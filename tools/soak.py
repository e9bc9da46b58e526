#!/usr/bin/env python3
"""Soak: interleaves every host-facing entry point many times and checks that device memory does not grow and
that results stay bit-identical (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
pdict, unitdict, _, xdict = problem.make_problem("mixed-6x64")
prob = con_dynamics.problem_arrays(pdict, unitdict); S = pdict["num_sections"]; ps = pdict["ps_params"]
E = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)])
x0 = pack_x(xdict)
X = problem.synthetic_batch(x0, E.M, 200)
for kind in ("alpha", "q", "qalpha"):
    E.aero_configure(kind, [(i, 1, 0.2 if kind != "q" else 4e4) for i in range(S - 1)])
r0, v0, _ = E.eval(x0)
rb0, jb0, _ = E.eval_batch(X)
J0, _ = E.jac_fd("vel", x0)
a0 = E.eval_aero("qalpha", x0)[0]
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
n = int(os.environ.get("SOAK_N", "300"))
for it in range(n):
    for _ in range(50):
        r, v, rc = E.eval(x0, out=v0.copy())
        assert rc == 0 and np.array_equal(r, r0)
    assert np.array_equal(v, v0)
    rb, jb, rc = E.eval_batch(X)
    assert rc == 0 and np.array_equal(rb, rb0) and np.array_equal(jb, jb0)
    if it % 10 == 0:
        J, _ = E.jac_fd("vel", x0 if it % 20 else x0 * (1 + 1e-12))
        if it % 20:
            assert np.array_equal(J, J0)
        assert np.array_equal(E.eval_aero("qalpha", x0)[0], a0)
        xb = x0.copy(); xb[3] = np.nan
        assert E.eval(xb)[2] == 1 and E.eval(x0)[2] == 0
free1 = torch.cuda.mem_get_info()[0]
print("soak ok: %d rounds, %.1f s, device memory drift %.1f MB" % (n, time.time() - t0, (free0 - free1) / 2**20))
assert abs(free0 - free1) < 64 * 2**20
