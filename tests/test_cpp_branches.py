"""Branches of the reference's C++ that its Python twins lack (so the goldens captured from the twins cannot pin them), against
known answers evaluated in 40-digit arithmetic (tests/golden/make_cpp_branches.py): the under-ground clamp of
src/gravity.cpp:45-47 and interp (src/wrapper_utils.hpp:51-80) around the rows of its table -- plus the quaternion
primitives of src/wrapper_coordinate.hpp:50-78 through their own point hooks.  CPU: the oracle; -m gpu: the
device functions through gel_point_eval."""
import numpy as np
import pytest

from conftest import load_golden

BARC20 = -0.484165371736e-3


def close(a, b, rtol, atol, what):
    err = np.abs(np.asarray(a) - np.asarray(b)) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), "%s: max excess %g" % (what, err.max())


def test_oracle_gravity_clamp_and_interp_rows_vs_exact():
    import oracle
    g = load_golden("g16_cpp_branches.npz")
    got = np.array([oracle.gravity(p, BARC20) for p in g["grav_pos"]])
    close(got, g["grav"], 4e-15, 1e-18, "oracle gravity (under-ground clamp included)")
    Rb = 6378137.0 * (1.0 - 1.0 / 298.257223563)
    under = np.linalg.norm(g["grav_pos"], axis=1) < Rb
    assert under.sum() >= 30
    # the clamp is there: without it a point at 5 % of the radius would feel 400 times the surface gravity
    assert np.all(np.linalg.norm(g["grav"][under], axis=1) < 1.02 * 3.986004418e14 / Rb ** 2)
    ys = np.array([oracle.interp(x, g["interp_xp"], g["interp_yp"]) for x in g["interp_x"]])
    close(ys, g["interp_y"], 4e-16, 0.0, "oracle interp")
    # x == xp[0]: np.interp's yp[0] (the C++ expression indexes xp[-1] there: no defined value to reproduce)
    assert oracle.interp(g["interp_xp"][0], g["interp_xp"], g["interp_yp"]) == g["interp_yp"][0]


@pytest.mark.gpu
def test_device_gravity_clamp_and_interp_rows_vs_exact():
    from gelato_amd.dynamics import point_eval
    g = load_golden("g16_cpp_branches.npz")
    got = point_eval(2, g["grav_pos"], aux=np.array([BARC20]))
    close(got, g["grav"], 8e-15, 1e-18, "device gravity (under-ground clamp included)")
    tab = np.column_stack([g["interp_xp"], g["interp_yp"]])
    ys = point_eval(6, g["interp_x"], aux=tab)[:, 0]
    close(ys, g["interp_y"], 1e-15, 0.0, "device interp")     # y_l + (x - x_l) * slope with the tabulated slope: <= 2 ulp
    assert point_eval(6, g["interp_xp"][:1], aux=tab)[0, 0] == g["interp_yp"][0]


@pytest.mark.gpu
def test_device_quaternion_primitives_vs_reference_golden_and_oracle():
    """quatmult, conj, quatrot on their own: kinds 11-13 of gel_point_eval against the values the reference's own
    functions returned (g2_g5: fr_quatmult, fr_quatrot_x) and the oracle, on unit AND non-unit quaternions"""
    import oracle
    from gelato_amd.dynamics import point_eval
    g = load_golden("g2_g5_pointwise.npz")
    q = g["fr_quat"]
    n = len(q)
    p = q[(np.arange(n) + 7) % n]
    qm = point_eval(11, np.column_stack([q, p]))
    close(qm, g["fr_quatmult"], 1e-15, 1e-16, "quatmult vs reference")
    c13 = point_eval(13, q)
    assert np.array_equal(c13[:, :4], q * [1.0, -1.0, -1.0, -1.0])                      # conj: sign flips, exact
    close(c13[:, 4:], g["fr_quatrot_x"], 1e-15, 2e-16, "thrust direction = quatrot(conj(q), ex) vs reference")
    ex = np.tile([1.0, 0.0, 0.0], (n, 1))
    qr = point_eval(12, np.column_stack([q * [1.0, -1.0, -1.0, -1.0], ex]))
    close(qr, g["fr_quatrot_x"], 1e-15, 2e-16, "quatrot vs reference")
    rng = np.random.default_rng(5)
    Q, V = rng.standard_normal((200, 4)) * 3.0, rng.standard_normal((200, 3)) * 1e4    # not normalised: the functions are algebraic
    close(point_eval(12, np.column_stack([Q, V])), np.array([oracle.quatrot(a, b) for a, b in zip(Q, V)]), 1e-14, 1e-9, "quatrot vs oracle")
    P2 = rng.standard_normal((200, 4))
    close(point_eval(11, np.column_stack([Q, P2])), np.array([oracle.quatmult(a, b) for a, b in zip(Q, P2)]), 1e-14, 1e-14, "quatmult vs oracle")
