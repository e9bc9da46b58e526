#!/usr/bin/env python3
"""Max |GPU - oracle| of residuals and x-dependent Jacobian entries on the named workloads (GPU box)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from gelato_amd import Engine, con_dynamics, pack_x, problem
for name in sys.argv[1:] or ["example", "3x32", "mixed-6x64", "dense-6x64", "stress-12x128"]:
    pdict, unitdict, _, xdict = problem.make_problem(name)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    P = oracle.Problem(prob)
    D = [P.D(i) for i in range(P.S)]; tau = [P.tau(i) for i in range(P.S)]
    E = Engine(prob, D=D, tau=tau)
    X = problem.synthetic_batch(pack_x(xdict), E.M, 3)
    res, jv, rc = E.eval_batch(X)
    ores, ovals = P.eval_batch(X)
    full = E.expand(jv)
    vm = E.var_mask()
    d = np.abs(full - ovals)[:, vm]; ref = np.abs(ovals)[:, vm]
    excess = d - 1e-6 * ref
    print("%-14s residual %.2e   jac max|d| %.2e   worst (|d| - 1e-6|ref|) %.2e of 1e-5   max|ref| %.1f" %
          (name, np.abs(res - ores).max(), d.max(), excess.max(), ref.max()), flush=True)
