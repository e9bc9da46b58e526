"""End-to-end consumer of the callback surface (BASELINE.json configs[0], the plumbing config, on the GPU): damped
Gauss-Newton feasibility steps on the shipped example through ``driver.make_callbacks`` -- all 23 keys of
Trajectory_Optimization.py:194-312 assembled into one scipy.sparse matrix (tests/gn_consumer.py) -- must reduce the
constraint violation and follow the committed trace of the same loop driven by the CPU oracle
(tests/golden/g17_gn_trace.npz, tests/golden/make_gn_trace.py).  This is the only way short of IPOPT (absent here) to show
that the assembled Jacobian is right as a matrix, not just entry by entry."""
import numpy as np
import pytest

import gn_consumer
from conftest import load_golden


def test_oracle_driven_loop_reproduces_its_committed_trace():
    """CPU: guards the fixture (and the consumer) against drift"""
    from gelato_amd import problem
    g = load_golden("g17_gn_trace.npz")
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    objfunc, sens = gn_consumer.oracle_callbacks(pdict, unitdict, condition)
    tr = gn_consumer.gauss_newton(objfunc, sens, xdict, iterations=2)
    for k in range(3):
        assert abs(tr[k]["norm"] - g["norms"][k]) <= 1e-9 * g["norms"][k] + 1e-12
        assert np.max(np.abs(tr[k]["x"] - g["X"][k])) <= 1e-10
        assert tr[k]["rows"] == g["rows"][k] and tr[k]["nnz"] == g["nnz"][k]
    assert g["norms"][-1] < 1e-4 * g["norms"][0]                       # the loop the engine has to follow does converge


@pytest.mark.gpu
def test_engine_callbacks_drive_gauss_newton_like_the_oracle():
    from gelato_amd import con_dynamics, con_user, driver, problem
    from gelato_amd.examples import user_constraints as uc
    g = load_golden("g17_gn_trace.npz")
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    con_user.set_user_module(uc)
    try:
        objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
        tr = gn_consumer.gauss_newton(objfunc, sens, xdict)
    finally:
        con_user.set_user_module(None)
    norms = np.array([t["norm"] for t in tr])
    assert len(tr[0]["groups"]) == 18 and tr[0]["rows"] == g["rows"][0] and tr[0]["nnz"] == g["nnz"][0]
    # the violation contracts: four and a half orders of magnitude in six steps
    assert norms[-1] < 1e-4 * norms[0] and norms[1] < 0.1 * norms[0]
    # ... along the oracle-driven trace: same active sets, same iterates to what the finite-difference entries' agreement
    # (1e-6 relative) allows after being passed through six linear solves
    for k, t in enumerate(tr):
        assert t["rows"] == g["rows"][k] and t["nnz"] == g["nnz"][k], (k, t["rows"], g["rows"][k])
        assert abs(t["norm"] - g["norms"][k]) <= 1e-4 * g["norms"][k] + 1e-7, (k, t["norm"], g["norms"][k])
        assert np.max(np.abs(t["x"] - g["X"][k])) <= 2e-6, (k, np.max(np.abs(t["x"] - g["X"][k])))
    # the objective is linear in x: its change over a step is its gradient times the step
    for k in range(len(tr) - 1):
        assert abs((tr[k + 1]["obj"] - tr[k]["obj"]) - tr[k]["gdotdx"]) <= 1e-12
    assert con_dynamics.last_status(pdict) == 0
