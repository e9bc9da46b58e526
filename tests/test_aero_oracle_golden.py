"""Pins the oracle's aero path constraints (lib/con_aero.py,
src/wrapper_utils.hpp:89-206) against the fixture captured from the imported reference (G9). CPU only.

Tolerances: constraint values |d| <= 1e-12 + 1e-10*|ref|.  FD gradients: two fp64 implementations of the same quotient differ
by their rounding noise divided by dx -- angle of attack and q-alpha go through acos(c) with c -> 1 at small angles (one ulp of c
moves alpha by eps/sin(alpha)), the position sweeps difference the altitude's cancellation.  Each entry gets the bound that
follows from the arithmetic at ITS node (tests/fd_noise.py aero_bound; tests/test_aero_exact_fd.py holds the oracle, the
reference's values and the engine to it around exact-arithmetic quotients): here |oracle - reference| <= 2 bounds (the drift of
the reference's in-place perturbation is the same fp64 arithmetic on both sides)."""
# constraint values: one ulp of cos(alpha) is eps/sin(alpha) in alpha, times q/limit for q-alpha (3e-13 at
# q = 30 kPa, alpha = 1 deg) -- and the example trajectory rides the q-alpha limit (values ~1e-8)
CTOL = {"alpha": 1e-11, "q": 1e-12, "qalpha": 1e-11}
import numpy as np
import pytest

import fd_noise
import oracle
from conftest import D_tau_from_golden, load_golden, problem_from_golden

KINDS = ["alpha", "q", "qalpha"]
VARS = ["position", "velocity", "quaternion", "t"]


def spec_from_golden(g, cname, kind):
    s = g["%s_%s_spec" % (cname, kind)].reshape(-1, 3).copy()
    if kind in ("alpha", "qalpha") and len(s):
        s[:, 2] = s[:, 2] * np.pi / 180.0          # units[3] = value * pi / 180  (con_aero.py:119,232)
    return s


@pytest.mark.parametrize("cname", ["example", "synthetic"])
def test_g9_aero_constraints(cname):
    g = load_golden("g9_aero_example.npz")
    prob = problem_from_golden(g)
    D, tau = D_tau_from_golden(g, prob)
    P = oracle.Problem(prob, D=D, tau=tau)
    x = g["x"]
    for kind in KINDS:
        spec = spec_from_golden(g, cname, kind)
        P.aero_configure(kind, spec)
        con = P.aero_residual(kind, x)
        if len(spec) == 0:
            assert con.size == 0
            continue
        ref = g["%s_%s_con" % (cname, kind)]
        assert con.shape == ref.shape
        assert np.all(np.abs(con - ref) <= CTOL[kind] + 1e-10 * np.abs(ref)), (kind, np.abs(con - ref).max())
        J = P.aero_jacobian(kind, x)
        bounds = fd_noise.aero_coo_bounds(oracle, dict(prob, tau=tau), x, kind, spec)
        for var in VARS:
            key = "%s_%s_jac_%s" % (cname, kind, var)
            r, c, v = J[var]["coo"]
            assert np.array_equal(r, g[key + "_rows"]) and np.array_equal(c, g[key + "_cols"]), key
            assert tuple(g[key + "_shape"]) == J[var]["shape"], key
            rv = g[key + "_vals"]
            assert np.all(np.abs(v - rv) <= 2.0 * bounds[var] + 1e-9 * np.abs(rv)), (key, np.abs(v - rv).max())


def test_aero_known_answers():
    # zero wind, velocity along the body axis -> zero angle of attack; q = rho v^2 / 2 at sea level
    wind = np.array([[-1e8, 0, 0], [1e10, 0, 0]], dtype=float)
    L = oracle.lib()
    import ctypes as C
    dp = C.POINTER(C.c_double)
    L.orc_dynamic_pressure_pa.restype = C.c_double
    L.orc_angle_of_attack_all_rad.restype = C.c_double
    pos = np.array([6378137.0, 0.0, 0.0])
    vel = np.array([0.0, 7.2921151467e-5 * 6378137.0 + 100.0, 0.0])   # 100 m/s eastward w.r.t. the ground
    q = L.orc_dynamic_pressure_pa(pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), C.c_double(0.0),
                                  wind.ctypes.data_as(dp), C.c_int(2))
    assert abs(q - 0.5 * 1.225 * 100.0 ** 2) < 0.5
    quat = np.array([np.cos(np.pi / 4), 0.0, 0.0, np.sin(np.pi / 4)])  # body x -> ECI y
    a = L.orc_angle_of_attack_all_rad(pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), quat.ctypes.data_as(dp),
                                      C.c_double(0.0), wind.ctypes.data_as(dp), C.c_int(2))
    assert abs(a) < 1e-7
