"""Pins the CPU oracle (oracle/) against the fixtures captured from the imported
reference (tests/golden/make_golden.py).  CPU only.

Tolerances (fp64, stated per SURVEY 8c):
  * point functions / RHS / residuals:  |d| <= 1e-12 + 1e-10*|ref|  (J2 constant matched to the twin)
  * D: 1e-11 relative to the row max;  tau: 1e-14
  * FD Jacobian x-dependent entries:  |d| <= 1e-5 + 1e-6*|ref|  (FD noise floor: 1e-16/dx amplified)
  * constant Jacobian entries (D, +-1, 0): D tolerance
"""
import hashlib

import numpy as np
import pytest

import oracle
from conftest import D_tau_from_golden, load_golden, problem_from_golden

TW = oracle.BARC20_PY_TWIN


def close(a, b, rtol=1e-10, atol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), "max excess %g (max abs diff %g)" % (err.max(), np.abs(a - b).max())


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ---------------- G1: LGR nodes and differentiation matrices ----------------
@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 8, 16, 32, 64, 128])
def test_g1_lgr(n):
    g = load_golden("g1_lgr.npz")
    tau, D = oracle.lgr_nodes(n), oracle.lgr_diffmat(n)
    assert np.max(np.abs(tau - g["tau_%d" % n])) <= 1e-14
    assert tau[-1] == 1.0
    Dg = g["D_%d" % n]
    rowmax = np.max(np.abs(Dg), axis=1, keepdims=True)
    assert np.max(np.abs(D - Dg) / rowmax) <= 1e-11
    # internal consistency the reference itself satisfies
    assert np.max(np.abs(D.sum(axis=1))) <= 1e-10 * np.abs(D).max()


# ---------------- G2: atmosphere ----------------
def test_g2_atmosphere():
    g = load_golden("g2_g5_pointwise.npz")
    alt = g["air_alt"]
    close([oracle.geopotential_altitude(z) for z in alt], g["air_geopot"])
    close([oracle.air_temperature(z) for z in alt], g["air_T"])
    close([oracle.air_pressure(z) for z in alt], g["air_P"])
    close([oracle.air_density(z) for z in alt], g["air_rho"])
    close([oracle.speed_of_sound(z) for z in alt], g["air_a"])


def test_g2_known_answers():
    # US-1976 layer-base values are the table itself (src/Air.cpp:31-42)
    assert oracle.air_temperature(0.0) == 288.15
    assert oracle.air_pressure(0.0) == 101325.0
    assert abs(oracle.air_pressure(11000.0) - 22632.0) < 1e-9
    assert abs(oracle.air_density(0.0) - 1.225) < 1e-4
    assert abs(oracle.speed_of_sound(0.0) - 340.294) < 1e-3


# ---------------- G3: frames / gravity ----------------
def test_g3_frames():
    g = load_golden("g2_g5_pointwise.npz")
    pos, vel, t, q = g["fr_pos"], g["fr_vel"], g["fr_t"], g["fr_quat"]
    n = len(t)
    close([oracle.ecef2geodetic(*p) for p in pos], g["fr_geodetic"], atol=1e-9)  # alt in metres ~1e-9 abs
    close([oracle.gravity(p, TW) for p in pos], g["fr_gravity_twin"])
    close([oracle.quat_nedg2eci(p, tt) for p, tt in zip(pos, t)], g["fr_quat_nedg2eci"], atol=1e-15)
    close([oracle.vel_eci2ecef(v, p, tt) for v, p, tt in zip(vel, pos, t)], g["fr_vel_eci2ecef"], atol=1e-10)
    close([oracle.ecef2eci(v, tt) for v, tt in zip(vel, t)], g["fr_ecef2eci"], atol=1e-10)
    close([oracle.eci2ecef(p, tt) for p, tt in zip(pos, t)], g["fr_eci2ecef"], atol=1e-8)
    close([oracle.quatrot(q[i] * [1, -1, -1, -1], [1.0, 0, 0]) for i in range(n)], g["fr_quatrot_x"], atol=1e-15)
    close([oracle.quatmult(q[i], q[(i + 7) % n]) for i in range(n)], g["fr_quatmult"], atol=1e-15)
    # launch site known answer: first trajectory row
    lat, lon, alt = oracle.ecef2geodetic(*pos[0])
    assert abs(lat - 42.50587) < 1e-6 and abs(lon - 143.45659) < 1e-6 and abs(alt - 50.0) < 1e-3


def test_g3_gravity_cpp_constant_close_to_twin():
    # the production C++ constant differs from the twin's J2 by ~1.4e-9 relative
    g = load_golden("g2_g5_pointwise.npz")
    for p in g["fr_pos"][::10]:
        d = oracle.gravity(p, oracle.BARC20_CPP) - oracle.gravity(p, TW)
        assert 1e-10 < np.linalg.norm(d) < 5e-8


# ---------------- G4: interpolation ----------------
def test_g4_interp():
    g = load_golden("g2_g5_pointwise.npz")
    W, CA = g["prob_wind_table"], g["prob_ca_table"]
    close([oracle.wind_ned(a, W) for a in g["wind_alt"]], g["wind_ned"], atol=1e-12)
    close([oracle.interp(m, CA[:, 0], CA[:, 1]) for m in g["ca_mach"]], g["ca_val"], atol=1e-15)
    # the well-defined value at the table's first knot (appendix C-3)
    assert oracle.interp(0.0, CA[:, 0], CA[:, 1]) == CA[0, 1]


# ---------------- G5: node-batched RHS ----------------
def test_g5_rhs():
    g = load_golden("g2_g5_pointwise.npz")
    prob = problem_from_golden(g)
    P = oracle.Problem(prob, barC20=TW)
    X = P.split_x(g["rhs_x"])
    units = prob["units"][:3]
    rv, rn, rq = [], [], []
    xa = ua = 0
    for i, n in enumerate(prob["num_nodes"]):
        xb, ub = xa + n + 1, ua + n
        param = np.array([prob["thrust"][i], prob["massflow"][i], prob["reference_area"][i], 0, prob["nozzle_area"][i]])
        pa = param.copy()
        if pa[2] == 0.0:
            pa[2] = 2.21
        m_ = X["mass"][xa:xb]
        p_ = X["position"].reshape(-1, 3)[xa:xb]
        v_ = X["velocity"].reshape(-1, 3)[xa:xb]
        q_ = X["quaternion"].reshape(-1, 4)[xa:xb]
        u_ = X["u"].reshape(-1, 2)[ua:ub]
        tn = np.concatenate([[X["t"][i]], P.tau(i) * (X["t"][i + 1] - X["t"][i]) / 2 + (X["t"][i + 1] + X["t"][i]) / 2])
        rv.append(oracle.dynamics_velocity(m_, p_, v_, q_, tn, pa, prob["wind_table"], prob["ca_table"], units, TW))
        rn.append(oracle.dynamics_velocity_NoAir(m_, p_, q_, param, units, TW))
        rq.append(oracle.dynamics_quaternion(q_[1:], u_, prob["units"][3]))
        xa, ua = xb, ub
    close(np.concatenate(rv), g["rhs_vel_air"])
    close(np.concatenate(rn), g["rhs_vel_noair"])
    close(np.concatenate(rq), g["rhs_quat"], atol=1e-16)


# ---------------- G6: residuals and COO Jacobians ----------------
def _variable_mask(P, prob, group, var, rows, cols):
    """True where the entry depends on x (so FD noise applies)."""
    if (group, var) in [("mass", "mass"), ("mass", "t"), ("pos", "position")]:
        return np.zeros(len(rows), dtype=bool)
    if (group, var) in [("vel", "velocity"), ("quat", "quaternion")]:
        k = 3 if group == "vel" else 4
        keep = np.zeros(len(rows), dtype=bool)
        ua = xa = 0
        for i, n in enumerate(prob["num_nodes"]):
            m = (rows // k >= ua) & (rows // k < ua + n)
            keep |= m & ((cols // k) - xa == (rows // k) - ua + 1)
            ua += n
            xa += n + 1
        return keep
    return np.ones(len(rows), dtype=bool)


@pytest.mark.parametrize("name", ["example", "3x32", "mixed6x64", "dense6x64", "negarea"])
@pytest.mark.parametrize("own_lgr", [False, True])
def test_g6_residuals_and_jacobians(name, own_lgr):
    g = load_golden("g6_%s.npz" % name)
    prob = problem_from_golden(g)
    if own_lgr:
        P = oracle.Problem(prob, barC20=TW)           # oracle's own tau/D
        res_tol = dict(rtol=1e-10, atol=5e-11)          # D differs from the reference's by <=1e-11*rowmax
    else:
        D, tau = D_tau_from_golden(g, prob)
        P = oracle.Problem(prob, barC20=TW, D=D, tau=tau)
        res_tol = dict(rtol=1e-10, atol=1e-12)
    x = g["x"]
    assert x.size == P.nvars
    for grp in oracle.GROUPS:
        close(P.residual(grp, x), g["res_" + grp], **res_tol)
        J = P.jacobian(grp, x)
        for var in oracle.BLOCK_VARS[grp]:
            key = "jac_%s_%s" % (grp, var)
            r, c, v = J[var]["coo"]
            assert r.dtype == np.int32 and c.dtype == np.int32 and v.dtype == np.float64
            assert tuple(g[key + "_shape"]) == J[var]["shape"]
            assert int(g[key + "_nnz"]) == len(v)
            assert str(g[key + "_rows_sha"]) == sha(r), key
            assert str(g[key + "_cols_sha"]) == sha(c), key
            if key + "_rows" in g:
                assert np.array_equal(r, g[key + "_rows"]) and np.array_equal(c, g[key + "_cols"])
            vm = _variable_mask(P, prob, grp, var, r, c)
            if key + "_vals" in g:
                ref = g[key + "_vals"]
                d = np.abs(v - ref)
                assert np.all(d[vm] <= 1e-5 + 1e-6 * np.abs(ref[vm])), (key, d[vm].max())
                if (~vm).any():
                    scale = max(1.0, np.abs(ref[~vm]).max())
                    assert d[~vm].max() <= 1e-11 * scale, (key, d[~vm].max())
            else:  # mixed6x64: x-dependent entries + sums only
                if "var_%s_%s_idx" % (grp, var) in g:
                    idx = g["var_%s_%s_idx" % (grp, var)]
                    ref = g["var_%s_%s_vals" % (grp, var)]
                    assert np.array_equal(idx, np.nonzero(vm)[0])
                    d = np.abs(v[idx] - ref)
                    assert np.all(d <= 1e-5 + 1e-6 * np.abs(ref)), (key, d.max())
                assert abs(v.sum() - float(g[key + "_vals_sum"])) <= 1e-5 * max(1, vm.sum()) + 1e-9 * len(v)


def test_g6_nnz_counts_match_survey():
    g = load_golden("g6_example.npz")
    P = oracle.Problem(problem_from_golden(g))
    assert P.block_nnz == {"mass": [586, 114], "pos": [1794, 198, 396], "vel": [198, 594, 5382, 792, 396],
                           "quat": [8720, 416, 416]}
    g = load_golden("g6_mixed6x64.npz")
    P = oracle.Problem(problem_from_golden(g))
    assert P.total_nnz == 607424 and P.nvars == 5065


# ---------------- G7: generic dense forward difference ----------------
def test_g7_jac_fd():
    g6 = load_golden("g6_example.npz")
    g7 = load_golden("g7_jacfd_example.npz")
    prob = problem_from_golden(g6)
    D, tau = D_tau_from_golden(g6, prob)
    P = oracle.Problem(prob, barC20=TW, D=D, tau=tau)
    x = g6["x"]
    for grp in oracle.GROUPS:
        J = P.jac_fd(grp, x)
        Xs = P.split_x(np.arange(P.nvars))
        for k, idx in Xs.items():
            ref = g7["%s_%s" % (grp, k)]
            d = np.abs(J[:, idx.astype(int)] - ref)
            assert np.all(d <= 1e-5 + 1e-6 * np.abs(ref)), (grp, k, d.max())
        # cross-check: structured COO vs generic FD agree to the FD noise floor
        dense = np.zeros_like(J)
        off = {"mass": 0, "position": P.M, "velocity": 4 * P.M, "quaternion": 7 * P.M, "u": 11 * P.M,
               "t": 11 * P.M + 2 * P.N}
        for var, blk in P.jacobian(grp, x).items():
            r, c, v = blk["coo"]
            dense[r, c + off[var]] += v
        assert np.max(np.abs(dense - J)) <= 1e-5


# ---------------- G8: objective ----------------
def test_g8_cost():
    g6 = load_golden("g6_example.npz")
    g8 = load_golden("g8_cost.npz")
    P = oracle.Problem(problem_from_golden(g6))
    x = g6["x"]
    assert P.cost(x, True) == float(g8["cost_Payload"])
    assert P.cost(x, False) == float(g8["cost_Other"])
    assert np.array_equal(P.cost_jac(x, True), g8["costjac_Payload_mass"])
    assert np.array_equal(P.cost_jac(x, False), g8["costjac_Other_t"])
