"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Run with `-m gpu` on an MI355X.

Tolerances (fp64):
  point functions / RHS / residuals : |d| <= 1e-12 + 1e-10*|ref|   (GPU libm (ocml) vs glibc: 1-2 ulp)
  Jacobian, x-dependent entries     : |d| <= 1e-5  + 1e-6*|ref|    (FD with dx = 1e-8 amplifies ulps by 1e8)
  Jacobian, constant entries        : bit-exact (same D handed to both sides)
  sparsity pattern (int32)          : bit-exact
"""
import hashlib

import numpy as np
import pytest

from conftest import D_tau_from_golden, load_golden, problem_from_golden

pytestmark = pytest.mark.gpu

TW = None


def _setup():
    global TW
    import oracle
    TW = oracle.BARC20_PY_TWIN
    return oracle


def close(a, b, rtol=1e-10, atol=1e-12, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert np.all(err <= 0), "%s: max excess %g (max abs diff %g)" % (what, np.nanmax(err), np.nanmax(np.abs(a - b)))


def dx_roundoff_bound(E, x):
    """Rigorous bound of the summation-order / FMA round-off of the D.X rows (NumPy's
    dot leaves the order unspecified): (n+1) * eps * sum_i |D[j,i]| * |X[i,c]| per residual row."""
    eps = np.finfo(np.float64).eps
    X = E.split_x(x)
    cols = {"mass": (X["mass"].reshape(-1, 1), 1), "pos": (X["position"].reshape(-1, 3), 3),
            "vel": (X["velocity"].reshape(-1, 3), 3), "quat": (X["quaternion"].reshape(-1, 4), 4)}
    out = {g: np.zeros(E.N * k) for g, (_, k) in cols.items()}
    ua = xa = 0
    for i, n in enumerate(E.num_nodes):
        n = int(n)
        aD = np.abs(E.D(i))
        for g, (arr, k) in cols.items():
            b = (n + 1) * eps * (aD @ np.abs(arr[xa:xa + n + 1]))
            out[g][ua * k:(ua + n) * k] = b.ravel()
        ua += n
        xa += n + 1
    return out


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_pair(prob, D=None, tau=None, barC20=None, flags=0):
    """(engine, oracle problem) on the same static problem and the same D / tau."""
    oracle = _setup()
    from gelato_amd import Engine
    bc = oracle.BARC20_CPP if barC20 is None else barC20
    P = oracle.Problem(prob, barC20=bc, D=D, tau=tau)
    if D is None:
        D = [P.D(i) for i in range(P.S)]
        tau = [P.tau(i) for i in range(P.S)]
    E = Engine(prob, D=D, tau=tau, barC20=bc, flags=flags)
    return E, P


def named_problem(name):
    from gelato_amd import con_dynamics, pack_x, problem
    pdict, unitdict, condition, xdict = problem.make_problem(name)
    return con_dynamics.problem_arrays(pdict, unitdict), pack_x(xdict), pdict


# --------------------------------------------------------------------------
# a4..a9: point functions
# --------------------------------------------------------------------------
def test_point_functions_vs_oracle_and_golden():
    oracle = _setup()
    from gelato_amd.dynamics import point_eval
    g = load_golden("g2_g5_pointwise.npz")
    alt = g["air_alt"]
    out = point_eval(0, alt)
    ref = np.array([[oracle.geopotential_altitude(z)] + [f(oracle.geopotential_altitude(z)) for f in
                    (oracle.air_temperature, oracle.air_pressure, oracle.air_density, oracle.speed_of_sound)]
                    for z in alt])
    close(out, ref, what="atmosphere vs oracle")
    # golden columns were evaluated at `alt` taken as the already-geopotential argument
    out_g = point_eval(0, alt)  # same kernel; compare T/P/rho/a at h = geopot(alt) with the oracle only
    assert np.array_equal(out, out_g)

    pos, vel, t = g["fr_pos"], g["fr_vel"], g["fr_t"]
    close(point_eval(1, pos), g["fr_geodetic"], atol=1e-9, what="ecef2geodetic")
    close(point_eval(2, pos, aux=np.array([TW])), g["fr_gravity_twin"], what="gravity")
    close(point_eval(2, pos, aux=np.array([oracle.BARC20_CPP])),
          np.array([oracle.gravity(p, oracle.BARC20_CPP) for p in pos]), what="gravity cpp const")
    wned = np.column_stack([np.linspace(-30, 30, len(t)), np.linspace(25, -5, len(t))])
    ref_w = np.array([oracle.quatrot(q, [a, b, 0.0]) for q, (a, b) in zip(g["fr_quat_nedg2eci"], wned)])
    close(point_eval(3, np.column_stack([pos, t, wned])), ref_w, atol=1e-13, what="wind NED->ECI via quat_nedg2eci")
    close(point_eval(4, np.column_stack([vel, pos, t])), g["fr_vel_eci2ecef"], atol=1e-10, what="vel_eci2ecef")
    W, CA = g["prob_wind_table"], g["prob_ca_table"]
    close(point_eval(5, g["wind_alt"], aux=W), g["wind_ned"], atol=1e-12, what="wind_ned")
    close(point_eval(6, g["ca_mach"], aux=CA).ravel(), g["ca_val"], atol=1e-15, what="interp CA")
    assert point_eval(6, np.array([0.0]), aux=CA)[0, 0] == CA[0, 1]      # appendix C-3: np.interp value at xp[0]


def test_atmosphere_dense_sweep_vs_oracle():
    oracle = _setup()
    from gelato_amd.dynamics import point_eval
    alt = np.concatenate([np.linspace(-500.0, 130e3, 4001), np.linspace(130e3, 900e3, 500)])
    out = point_eval(0, alt)
    h = np.array([oracle.geopotential_altitude(z) for z in alt])
    ref = np.column_stack([h, [oracle.air_temperature(z) for z in h], [oracle.air_pressure(z) for z in h],
                           [oracle.air_density(z) for z in h], [oracle.speed_of_sound(z) for z in h]])
    close(out, ref, what="atmosphere sweep")


# --------------------------------------------------------------------------
# a1..a3: node-batched RHS
# --------------------------------------------------------------------------
def test_rhs_functions_vs_golden_and_oracle():
    oracle = _setup()
    from gelato_amd import dynamics
    g = load_golden("g2_g5_pointwise.npz")
    prob = problem_from_golden(g)
    P = oracle.Problem(prob, barC20=TW)
    X = P.split_x(g["rhs_x"])
    units = prob["units"][:3]
    rv, rn, rq, ov = [], [], [], []
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
        rv.append(dynamics.dynamics_velocity(m_, p_, v_, q_, tn, pa, prob["wind_table"], prob["ca_table"], units, TW))
        ov.append(oracle.dynamics_velocity(m_, p_, v_, q_, tn, pa, prob["wind_table"], prob["ca_table"], units, TW))
        rn.append(dynamics.dynamics_velocity_NoAir(m_, p_, q_, param, units, TW))
        rq.append(dynamics.dynamics_quaternion(q_[1:], u_, prob["units"][3]))
        xa, ua = xb, ub
    close(np.concatenate(rv), np.concatenate(ov), what="dynamics_velocity vs oracle")
    close(np.concatenate(rv), g["rhs_vel_air"], what="dynamics_velocity vs golden")
    close(np.concatenate(rn), g["rhs_vel_noair"], what="dynamics_velocity_NoAir vs golden")
    close(np.concatenate(rq), g["rhs_quat"], atol=1e-16, what="dynamics_quaternion vs golden")


def test_parity_corners_underground_polar_and_air_at_rest():
    """Corners the synthetic trajectories never visit, GPU against the oracle (which follows the C++ sources):
    a node below the polar radius (the r < b clamp of src/gravity.cpp:45-47, a branch the Python twin lacks), a node
    exactly on the polar axis (p = 0: the reference's atan2(0, 0) = 0 longitude, finite everywhere), and a vehicle at
    rest in the air (air-relative speed exactly 0 -> Mach 0, the x == xp[0] corner of interp)."""
    oracle = _setup()
    from gelato_amd import dynamics
    from gelato_amd.dynamics import point_eval
    g = load_golden("g2_g5_pointwise.npz")
    prob = problem_from_golden(g)
    W, CA = prob["wind_table"], prob["ca_table"]
    units = prob["units"][:3]
    Rb = 6378137.0 * (1.0 - 1.0 / 298.257223563)
    rng = np.random.default_rng(3)
    d = rng.standard_normal((64, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    radii = np.concatenate([np.linspace(0.3, 0.999, 40), [1.0, 1.0 + 1e-12, 1.0001, 1.1] * 6]) * Rb
    pos = d * radii[:, None]
    got = point_eval(2, pos, aux=np.array([oracle.BARC20_CPP]))
    ref = np.array([oracle.gravity(p, oracle.BARC20_CPP) for p in pos])
    close(got, ref, what="gravity with nodes below the polar radius")
    inside = radii < Rb
    # under ground the magnitude no longer grows with 1/r^2: the clamp holds the radial scale at r = b
    assert np.all(np.linalg.norm(got[inside], axis=1) < 1.02 * 3.986004418e14 / Rb ** 2)
    # RHS at an underground node, on the polar axis (north and south), and at rest in the air (zero wind table, v = omega x r)
    W0 = W.copy()
    W0[:, 1:] = 0.0
    omega = 7.2921151467e-5
    r_rest = np.array([[4.2e6, 3.9e6, 2.9e6], [-5.0e6, 1.0e6, 4.0e6]]) * 1.02
    v_rest = np.column_stack([-(omega * r_rest[:, 1]), omega * r_rest[:, 0], np.zeros(2)])
    cases = {
        "underground": (pos[:8] * 1.0, rng.standard_normal((8, 3)) * 300.0, W),
        "polar": (np.array([[0.0, 0.0, 6.45e6], [0.0, 0.0, -6.5e6], [0.0, 0.0, 6.36e6]]), rng.standard_normal((3, 3)) * 500.0, W),
        "at rest": (r_rest, v_rest, W0),
    }
    param = np.array([4.2e5, 140.0, 2.21, 0.0, 0.68])
    for name, (r, v, wt) in cases.items():
        n = len(r)
        m = np.full(n, 0.7)
        q = rng.standard_normal((n, 4))
        q /= np.linalg.norm(q, axis=1)[:, None]
        t = np.linspace(0.0, 0.9, n)
        a_gpu = dynamics.dynamics_velocity(m, r / units[1], v / units[2], q, t, param, wt, CA, units)
        a_ref = oracle.dynamics_velocity(m, r / units[1], v / units[2], q, t, param, wt, CA, units)
        assert np.all(np.isfinite(a_gpu)), name
        close(a_gpu, a_ref, rtol=1e-9, atol=1e-11, what="RHS corner: " + name)
    # at rest in the air the aerodynamic force is exactly zero: the same acceleration as a phase without aerodynamics
    n = len(r_rest)
    m = np.full(n, 0.7)
    q = np.tile([0.5, 0.5, -0.5, 0.5], (n, 1))
    a_air = dynamics.dynamics_velocity(m, r_rest / units[1], v_rest / units[2], q, np.zeros(n), np.array([4.2e5, 140.0, 2.21, 0.0, 0.0]),
                                       W0, CA, units)
    a_no = dynamics.dynamics_velocity_NoAir(m, r_rest / units[1], q, np.array([4.2e5, 140.0, 0.0, 0.0, 0.0]), units)
    close(a_air, a_no, rtol=1e-13, atol=1e-15, what="zero air-relative speed: no aerodynamic force")


def test_engine_generated_lgr_on_the_device_vs_reference_golden():
    """The engine's OWN nodes and differentiation matrices (gel_lgr_nodes / gel_lgr_diffmat, Newton + barycentric
    weights on the host) on the device, against the reference's residuals and Jacobians of the example problem, whose
    D came from the reference's lib/PSfunctions.py.  D differs by <= 1e-11 of a row maximum (G1), so the residuals carry
    that times sum |X|; the x-dependent Jacobian entries do not see D at all except through D[j][j+1]."""
    oracle = _setup()
    from gelato_amd import Engine
    g = load_golden("g6_example.npz")
    prob = problem_from_golden(g)
    Dg, taug = D_tau_from_golden(g, prob)
    E = Engine(prob, barC20=TW)                        # no D / tau handed over: the engine generates them
    for i, (Dr, tr) in enumerate(zip(Dg, taug)):
        assert np.max(np.abs(E.D(i) - Dr)) <= 1e-11 * np.max(np.abs(Dr)) and np.max(np.abs(E.tau(i) - tr)) <= 1e-14
    x = g["x"]
    res, vals, rc = E.eval(x)
    assert rc == 0
    R, J = E.split_res(res), E.jac_dicts(vals)
    X = E.split_x(x)
    scale = {"mass": np.abs(X["mass"]).max(), "pos": np.abs(X["position"]).max(), "vel": np.abs(X["velocity"]).max(),
             "quat": np.abs(X["quaternion"]).max()}
    vm = E.var_mask()
    b = 0
    for grp in oracle.GROUPS:
        n_max = int(max(prob["num_nodes"]))
        tol = 1e-11 * max(np.max(np.abs(d)) for d in Dg) * (n_max + 1) * scale[grp]
        assert np.all(np.abs(R[grp] - g["res_" + grp]) <= 1e-12 + 1e-10 * np.abs(g["res_" + grp]) + tol), grp
        for var in oracle.BLOCK_VARS[grp]:
            key = "jac_%s_%s" % (grp, var)
            r, c, v = J[grp][var]["coo"]
            assert np.array_equal(r, g[key + "_rows"]) and np.array_equal(c, g[key + "_cols"])
            m = vm[E.block_off[b]:E.block_off[b + 1]]
            ref = g[key + "_vals"]
            assert np.all(np.abs(v[m] - ref[m]) <= 1e-5 + 1e-6 * np.abs(ref[m])), key
            assert np.all(np.abs(v[~m] - ref[~m]) <= 1e-11 * (1.0 + np.max(np.abs(ref)))), key     # D entries, +-1, 0
            b += 1


def test_rhs_shape_errors_like_pybind():
    from gelato_amd import dynamics
    with pytest.raises(TypeError):
        dynamics.dynamics_quaternion(np.zeros((3, 4)), np.zeros((2, 2)), 1.0)
    assert dynamics.dynamics_quaternion(np.zeros((0, 4)), np.zeros((0, 2)), 1.0).shape == (0, 4)


# --------------------------------------------------------------------------
# a12..a19: residuals and COO Jacobians
# --------------------------------------------------------------------------
def reference_noise_per_row(P, prob, x, barC20=None):
    """What the REFERENCE's finite-difference entries of the velocity defect may be off the exact quotient, per residual row
    of the group (tests/fd_noise.py: derived from the arithmetic, checked against exact-arithmetic quotients in
    tests/test_exact_fd.py): {var: [3N]} -- zero for phases without aerodynamics; "lat_deg": the node's geodetic latitude
    (0 where there is no aerodynamics: nothing there needs an allowance)."""
    import fd_noise
    oracle = _setup()
    pr = dict(prob)
    pr["tau"] = [P.tau(i) for i in range(P.S)]
    terms = fd_noise.velocity_noise_terms(oracle, pr, x, barC20)
    N = int(np.sum(prob["num_nodes"]))
    pos, oth, lat = np.zeros(N), np.zeros(N), np.zeros(N)
    ua = 0
    for i, n in enumerate(prob["num_nodes"]):
        n = int(n)
        if terms[i] is not None:
            pos[ua:ua + n] = fd_noise.reference_bound(terms[i])
            oth[ua:ua + n] = fd_noise.reference_bound_other(terms[i])
            lat[ua:ua + n] = np.rad2deg(terms[i]["lat"])
        ua += n
    return {"position": np.repeat(pos, 3), "mass": np.repeat(oth, 3), "velocity": np.repeat(oth, 3),
            "quaternion": np.repeat(oth, 3), "t": np.repeat(oth, 3), "lat_deg": np.repeat(lat, 3)}


BENIGN_LAT_DEG = 55.0   # the latitude rounds 1-2 kept dense air below; the margins table reports the flat tolerance's excess below it separately


def defect_margins(E, P, x, prob=None, barC20=None):
    """Engine against oracle, every x-dependent Jacobian entry: per block the largest difference, how far it is inside the
    FLAT tolerance 1e-5 + 1e-6 |ref| of SURVEY 8(c), how many entries need the DERIVED allowance (the reference's own
    finite-difference noise at that node, velocity group of aerodynamic phases; only when `prob` is given), the largest
    allowance in use, and the worst flat excess among the rows below BENIGN_LAT_DEG.  -> (rows of the table, res, vals).
    tests/parity_margin.py writes the table to profiles/; check_against_oracle asserts on it."""
    oracle = _setup()
    res, rc = E.eval_residual(x)
    assert rc == 0
    vals, rc = E.eval_jacobian(x)
    assert rc == 0
    J = E.jac_dicts(vals)
    var_mask = E.var_mask()
    noise = reference_noise_per_row(P, prob, x, barC20) if prob is not None else None
    table, b = [], 0
    for grp in oracle.GROUPS:
        Jo = P.jacobian(grp, x)
        for var in oracle.BLOCK_VARS[grp]:
            r, c, v = J[grp][var]["coo"]
            ro, co, vo = Jo[var]["coo"]
            assert r.dtype == np.int32 and c.dtype == np.int32 and v.dtype == np.float64
            assert np.array_equal(r, ro) and np.array_equal(c, co), "pattern %s/%s" % (grp, var)
            assert J[grp][var]["shape"] == Jo[var]["shape"]
            m = var_mask[E.block_off[b]:E.block_off[b + 1]]
            assert np.array_equal(v[~m], vo[~m]), "%s/%s constants" % (grp, var)
            b += 1
            if not m.any():
                continue
            d = np.abs(v - vo)[m]
            flat = (1e-5 + 1e-6 * np.abs(vo))[m]
            allow = noise[var][r][m] if (noise is not None and grp == "vel") else np.zeros(d.shape)
            benign = (np.abs(noise["lat_deg"][r][m]) < BENIGN_LAT_DEG) if (noise is not None and grp == "vel") else np.ones(d.shape, dtype=bool)
            table.append({"block": "%s/%s" % (grp, var), "entries": int(d.size), "max_abs_diff": float(d.max()),
                          "max_abs_ref": float(np.abs(vo)[m].max()),
                          "worst_flat_excess": float((d - flat).max()),           # negative: inside the flat tolerance
                          "entries_needing_derived_allowance": int(np.count_nonzero(d > flat)),
                          "derived_allowance_max": float(allow.max()), "worst_total_excess": float((d - flat - allow).max()),
                          "benign_entries": int(benign.sum()),
                          "worst_flat_excess_benign": float((d - flat)[benign].max()) if benign.any() else None})
    return table, res, vals


def check_against_oracle(E, P, x, what, prob=None, barC20=None):
    """Engine against oracle: residuals 1e-12 + 1e-10 |ref| (+ the D.X summation bound), constants and pattern bit-exact,
    x-dependent Jacobian entries 1e-5 + 1e-6 |ref| -- plus, for the velocity group of aerodynamic phases when `prob` is given,
    the derived rounding noise of the reference's OWN finite differences at that node (high latitude, dense air: the
    oracle recomputes like the reference and carries that noise; the engine's exact-difference sweeps do not).  How many
    entries lean on the allowance is on the record (tests/parity_margin.py -> profiles/r04/parity_margins.json: a handful per
    extreme state); that the DEFAULT engine needs none of it against the exact quotients is asserted in tests/test_exact_fd.py."""
    oracle = _setup()
    table, res, vals = defect_margins(E, P, x, prob, barC20)
    R = E.split_res(res)
    bound = dx_roundoff_bound(E, x)
    for grp in oracle.GROUPS:
        close(R[grp], P.residual(grp, x), atol=1e-12 + bound[grp], what="%s residual %s" % (what, grp))
    for row in table:
        assert row["worst_total_excess"] <= 0.0, "%s %s var max %g (max excess %g)" % (what, row["block"], row["max_abs_diff"], row["worst_total_excess"])
    return res, vals


@pytest.mark.parametrize("flags", [0, 8])
@pytest.mark.parametrize("name", ["example", "3x32", "mixed6x64", "dense6x64", "negarea"])
def test_residuals_jacobians_vs_golden_and_oracle(name, flags):
    oracle = _setup()
    g = load_golden("g6_%s.npz" % name)
    prob = problem_from_golden(g)
    D, tau = D_tau_from_golden(g, prob)
    E, P = make_pair(prob, D, tau, barC20=TW, flags=flags)
    assert E.flags == flags
    x = g["x"]
    res, vals = check_against_oracle(E, P, x, name)
    R = E.split_res(res)
    J = E.jac_dicts(vals)
    for grp in oracle.GROUPS:
        close(R[grp], g["res_" + grp], what="golden residual " + grp)
        for var in oracle.BLOCK_VARS[grp]:
            key = "jac_%s_%s" % (grp, var)
            r, c, v = J[grp][var]["coo"]
            assert str(g[key + "_rows_sha"]) == sha(r) and str(g[key + "_cols_sha"]) == sha(c), key
            assert tuple(g[key + "_shape"]) == J[grp][var]["shape"]
            if key + "_vals" in g:
                ref = g[key + "_vals"]
                d = np.abs(v - ref)
                assert np.all(d <= 1e-5 + 1e-6 * np.abs(ref)), (key, d.max())
            elif "var_%s_%s_idx" % (grp, var) in g:
                idx, ref = g["var_%s_%s_idx" % (grp, var)], g["var_%s_%s_vals" % (grp, var)]
                d = np.abs(v[idx] - ref)
                assert np.all(d <= 1e-5 + 1e-6 * np.abs(ref)), (key, d.max())


@pytest.mark.parametrize("flags", [0, 8])
@pytest.mark.parametrize("name", ["dense-6x64", "mixed-6x64", "stress-12x128"])
def test_full_size_configs_vs_oracle(name, flags):
    prob, x, _ = named_problem(name)
    E, P = make_pair(prob, flags=flags)
    check_against_oracle(E, P, x, name)
    if name == "dense-6x64":
        assert E.total_nnz == 745728 and E.algorithmic_bytes == 320072
    if name == "stress-12x128":
        assert E.nvars == 20113 and E.nres == 16896


def test_ragged_and_edge_phases():
    """n = 2 phases, engine-off + free attitude, NoAir + hold, zero thrust with aero, and a
    multi-chunk phase (n = 100 > 64) with a ragged tail; positions all over the sphere (tests/states.py)."""
    import states
    prob, x = states.ragged_state()
    E, P = make_pair(prob)
    check_against_oracle(E, P, x, "ragged", prob=prob)


def test_dense_air_at_high_latitude():
    """55 .. 89.9 degrees, sea level .. 30 km, up to 2.5 km/s: the states rounds 1-2 kept out of the parity claim.  Against the
    oracle with the reference's derived noise allowance; against exact quotients in tests/test_exact_fd.py."""
    import states
    prob, x = states.polar_dense_state()
    E, P = make_pair(prob)
    check_against_oracle(E, P, x, "polar-dense", prob=prob)


def test_batch_matches_single_and_oracle():
    prob, x0, pdict = named_problem("3x32")
    from gelato_amd import problem
    E, P = make_pair(prob)
    X = problem.synthetic_batch(x0, E.M, 6)
    res, jv, rc = E.eval_batch(X)
    assert rc == 0 and res.shape == (6, E.nres) and jv.shape == (6, E.V)
    ores, ovals = P.eval_batch(X)
    close(res, ores, what="batch residual")
    full = E.expand(jv)
    vm = E.var_mask()
    d = np.abs(full - ovals)
    assert np.all(d[:, vm] <= 1e-5 + 1e-6 * np.abs(ovals[:, vm]))
    assert np.array_equal(full[:, ~vm], ovals[:, ~vm])
    # element b of a batch == the single-vector call, bit for bit
    for b in (0, 3, 5):
        r1, _ = E.eval_residual(X[b])
        v1, _ = E.eval_jacobian(X[b])
        assert np.array_equal(r1, res[b]) and np.array_equal(v1, full[b])
    # residual-only and fused launches agree bit for bit
    r_only, _, _ = E.eval_batch(X, want_jac=False)
    assert np.array_equal(r_only, res)


@pytest.mark.parametrize("name,Bs", [("example", (1, 2, 3, 5, 30, 31)), ("mixed-6x64", (45, 46, 47, 257)),
                                     # 32-node phases, two vectors per wavefront: odd batches leave a wavefront with ONE vector (its
                                     # residual rows then keep the strided store; full wavefronts write both vectors' rows through the tile)
                                     ("3x32", (1, 3, 7, 8, 9, 33)),
                                     ("stress-12x128", (11, 13, 37, 70))])   # 37, 70: 10 / 18 vector groups = one / two
                                                                             # blocks of eight plus a short one (launch order)
def test_cooperative_dx_form_equals_single_vector_calls(name, Bs):
    """Throughput launches form D.X per WORKGROUP (four decision vectors side by side on the matrix pipe, one row
    tile per wavefront); batches that are not a multiple of four leave wavefronts without a vector of their own.
    Every vector of such a batch must come out bit for bit as its single-vector call (split latency form)."""
    import torch
    from gelato_amd import Engine, problem
    prob, x0, _ = named_problem(name)
    E = Engine(prob, flags=1)   # GEL_FLAG_DX_MFMA: the matrix-pipe form also for the example's short phases
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    Bmax = max(Bs)
    X = problem.synthetic_batch(x0, E.M, Bmax, seed=77)
    singles = {}
    for B in Bs:
        info = E.launch_info(B)
        dX = torch.from_numpy(X[:B].copy()).to(dev)
        dres = torch.full((B + 1, E.nres), -7.0, dtype=torch.float64, device=dev)   # one guard row behind the batch
        djv = torch.full((B + 1, E.V), -7.0, dtype=torch.float64, device=dev)
        E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
        assert E.sync(s) == 0
        res, jv = dres.cpu().numpy(), djv.cpu().numpy()
        assert np.all(res[B] == -7.0) and np.all(jv[B] == -7.0), "a wavefront without a vector wrote something"
        # residual-only launches never split: always the cooperative form
        dres2 = torch.full((B + 1, E.nres), -7.0, dtype=torch.float64, device=dev)
        E.eval_batch_device(B, dX.data_ptr(), dres2.data_ptr(), 0, s)
        assert E.sync(s) == 0
        r2 = dres2.cpu().numpy()
        assert np.array_equal(r2[:B], res[:B]) and np.all(r2[B] == -7.0)
        for b in sorted({0, B // 2, B - 1}):
            if b not in singles:
                r1, v1, rc = E.eval(X[b])
                assert rc == 0
                singles[b] = (r1, v1[E.var_index()])
            assert np.array_equal(res[b], singles[b][0]), (B, b, info)
            assert np.array_equal(jv[b], singles[b][1]), (B, b, info)
    assert E.launch_info(Bmax, True, False)[2] == 0 and E.launch_info(1)[2] == 1


@pytest.mark.parametrize("name", ["3x32", "example", "3x16"])
def test_two_vectors_per_wavefront_equal_one_vector_per_wavefront(name):
    """Problems whose phases all have <= 32 nodes carry two decision vectors per wavefront (eight per workgroup) in the
    cooperative matrix-pipe form.  Same product, same order of accumulation: every batch size around the multiples of
    eight gives the bits of the one-vector-per-wavefront form (GEL_FLAG_NO_PACK), nothing is written behind the batch."""
    import torch
    from gelato_amd import Engine, problem
    prob, x0, _ = named_problem(name)
    Ep = Engine(prob, flags=1)        # matrix pipe; packed because every phase fits 32 lanes
    E1 = Engine(prob, flags=1 | 4)    # GEL_FLAG_NO_PACK
    assert Ep.launch_info(1024)[4] == 1 and E1.launch_info(1024)[4] == 0 and Ep.launch_info(64, True, False)[4] == 1
    assert Ep.launch_info(1024)[3] * 2 == E1.launch_info(1024)[3]
    assert Ep.launch_info(1)[2] == 1 and Ep.launch_info(1)[4] == 0            # the split latency form never packs
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    Bs = (7, 8, 9, 15, 16, 17, 100, 1001, 1003)
    X = problem.synthetic_batch(x0, Ep.M, max(Bs), seed=21)
    compared_jac = 0
    for B in Bs:
        dX = torch.from_numpy(X[:B].copy()).to(dev)
        split = bool(Ep.launch_info(B)[2])                                     # small batches: the fused launch splits instead
        out = []
        for E in (Ep, E1):
            dres = torch.full((B + 1, E.nres), -7.0, dtype=torch.float64, device=dev)
            djv = torch.full((B + 1, E.V), -7.0, dtype=torch.float64, device=dev)
            dres2 = torch.full((B + 1, E.nres), -7.0, dtype=torch.float64, device=dev)
            E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
            E.eval_batch_device(B, dX.data_ptr(), dres2.data_ptr(), 0, s)      # residual-only launches never split
            assert E.sync(s) == 0
            out.append((dres.cpu().numpy(), djv.cpu().numpy(), dres2.cpu().numpy()))
        for o in out:
            assert np.all(o[0][B] == -7.0) and np.all(o[1][B] == -7.0) and np.all(o[2][B] == -7.0), (B, "wrote behind the batch")
            assert np.array_equal(o[0], o[2])
        assert np.array_equal(out[0][2], out[1][2]), B
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), B
        compared_jac += not split
    assert compared_jac >= 2
    # a non-finite value in the second vector of a wavefront is seen
    Xb = X[:16].copy()
    Xb[5, Ep.M + 4] = np.nan
    dX = torch.from_numpy(Xb).to(dev)
    dres = torch.empty((16, Ep.nres), dtype=torch.float64, device=dev)
    Ep.eval_batch_device(16, dX.data_ptr(), dres.data_ptr(), 0, s)
    assert Ep.sync(s) == 1
    r = dres.cpu().numpy()
    assert np.isnan(r[5]).any() and np.isfinite(np.delete(r, 5, axis=0)).all()


def test_lookups_through_a_kept_interval_equal_fresh_lookups():
    """The fused kernel keeps the table interval of a node's previous CA / wind lookup and takes a short way when every
    lane of the wavefront falls into its own kept interval.  Sequences that sit on breakpoints, cross them by an ulp or by
    1e-8, leave the table at both ends, and carry NaN / inf -- different in every lane -- give the bits of fresh lookups."""
    from gelato_amd.dynamics import point_eval
    rng = np.random.default_rng(99)
    xp = np.array([0.0, 0.7, 1.0, 1.5, 2.0, 5.0, 100.0])
    tab = np.column_stack([xp, [0.3, 0.3, 0.65, 0.65, 0.6, 0.3, 0.3]])
    alt = np.array([-1e8, 0.0, 1e3, 3e3, 1.1e4, 1.5e4, 1.6e4, 2.3e4, 1e10])
    wind = np.column_stack([alt, 3.0 * np.sin(np.arange(9.0)), 10.0 * np.cos(np.arange(9.0))])
    wind[[0, 1, -1], 1:] = 0.0
    n = 64 * 40 + 17
    for kind_c, kind_f, t, cols in ((9, 6, tab, 1), (10, 5, wind, 3)):
        bp = t[:, 0]
        base = bp[rng.integers(0, len(bp), n)]
        seq = np.empty((n, 8))
        seq[:, 0] = base * (1.0 - 1e-9) - 1e-12                       # just below a breakpoint
        seq[:, 1] = base                                              # on it
        seq[:, 2] = np.nextafter(base, np.inf)                        # an ulp above
        seq[:, 3] = base * (1.0 + 1e-8) + 1e-8                        # a forward-difference step above
        seq[:, 4] = seq[:, 3] + 1e-8 * np.abs(seq[:, 3])              # and another (same interval: the short way)
        seq[:, 5] = rng.uniform(bp[0] - 1.0, min(bp[-1], 3e4) * 1.1, n)   # anywhere, also beyond both ends
        seq[:, 6] = seq[:, 5] * (1.0 + 1e-8)
        seq[:, 7] = seq[:, 6]
        calm = rng.random(n) < 0.5                                    # half of the lanes stay put: wavefronts with hits AND misses
        seq[calm, :] = seq[calm, 5:6] * (1.0 + 1e-9 * np.arange(8)[None, :])
        seq[rng.integers(0, n, 25), rng.integers(0, 8, 25)] = np.nan
        seq[rng.integers(0, n, 25), rng.integers(0, 8, 25)] = np.inf
        seq[rng.integers(0, n, 25), rng.integers(0, 8, 25)] = -np.inf
        got = point_eval(kind_c, seq, aux=t)
        fresh = point_eval(kind_f, seq.ravel(), aux=t)
        if kind_f == 5:
            fresh = fresh[:, :2].reshape(n, 16)
        else:
            fresh = fresh.reshape(n, 8)
        assert np.array_equal(got, fresh, equal_nan=True), (kind_c, np.argwhere(~((got == fresh) | (np.isnan(got) & np.isnan(fresh))))[:5])
        assert np.isnan(got).any() and np.isfinite(got).any()


@pytest.mark.parametrize("name,B,jac", [("mixed-6x64", 65536, True), ("dense-6x64", 65536, True), ("stress-12x128", 16384, True),
                                        ("3x32", 65536, False)])     # the batch sizes bench.py / tools/record_all.sh run
def test_full_size_batches_size_independent_properties(name, B, jac):
    """The bench's own launches (BASELINE.json configs at their full batch sizes), checked through properties that do not
    need a reference of that size: equal decision vectors give equal output rows wherever they sit in the batch (the batch
    tiles 256 distinct vectors), the reversed batch gives the reversed output, a second launch reproduces the first bit for
    bit, everything is finite, and sampled rows are the single-vector (split latency form) results."""
    import torch
    from gelato_amd import Engine, problem
    prob, x0, _ = named_problem(name)
    E = Engine(prob)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    P = 256
    Xd = problem.synthetic_batch(x0, E.M, P, seed=3)
    dXd = torch.from_numpy(Xd).to(dev)
    dX = dXd.repeat(B // P, 1).contiguous()
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((B, E.V), dtype=torch.float64, device=dev) if jac else None
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr() if jac else 0, s)
    assert E.sync(s) == 0
    outs = [dres] + ([djv] if jac else [])
    for o in outs:
        assert bool(torch.isfinite(o).all())
        t = o.view(B // P, P, -1)
        assert bool((t == t[0:1]).all()), "equal vectors, different rows"
    # determinism
    dres2 = torch.empty_like(dres)
    djv2 = torch.empty_like(djv) if jac else None
    E.eval_batch_device(B, dX.data_ptr(), dres2.data_ptr(), djv2.data_ptr() if jac else 0, s)
    assert E.sync(s) == 0 and torch.equal(dres2, dres) and (not jac or torch.equal(djv2, djv))
    # the reversed batch
    dXr = torch.flip(dX, dims=[0]).contiguous()
    E.eval_batch_device(B, dXr.data_ptr(), dres2.data_ptr(), djv2.data_ptr() if jac else 0, s)
    assert E.sync(s) == 0 and torch.equal(torch.flip(dres2, dims=[0]), dres) and (not jac or torch.equal(torch.flip(djv2, dims=[0]), djv))
    # sampled rows against single-vector calls
    for b in (0, 77, P - 1):
        if jac:
            r1, v1, rc = E.eval(Xd[b])
            assert rc == 0 and np.array_equal(dres[b].cpu().numpy(), r1) and np.array_equal(djv[B - P + b].cpu().numpy(), v1[E.var_index()])
        else:
            r1, rc = E.eval_residual(Xd[b])
            assert rc == 0 and np.array_equal(dres[B - P + b].cpu().numpy(), r1)


def test_device_pointer_api_and_full_expansion():
    import torch
    prob, x0, _ = named_problem("mixed-6x64")
    from gelato_amd import problem
    E, P = make_pair(prob)
    B = 5
    X = problem.synthetic_batch(x0, E.M, B)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    dfull = torch.empty((B, E.total_nnz), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    E.expand_full_device(B, djv.data_ptr(), dfull.data_ptr(), s)
    assert E.sync(s) == 0
    res, jv, _ = E.eval_batch(X)
    assert np.array_equal(dres.cpu().numpy(), res) and np.array_equal(djv.cpu().numpy(), jv)
    assert np.array_equal(dfull.cpu().numpy(), E.expand(jv))
    # the update-in-place mode: constants laid down once, then only the x-dependent entries per evaluation --
    # the same bits as the full rewrite, also after the buffer has served another batch, for one vector and for a ragged batch
    prob_e, x0_e, _ = named_problem("example")      # total_nnz = 20002: not a multiple of 8 -> the entry-wise update kernel
    E_e, _ = make_pair(prob_e)
    assert E.total_nnz % 8 == 0 and E_e.total_nnz % 8 != 0
    for E, x0, Bu in ((E, x0, 1), (E, x0, B), (E, x0, 37), (E_e, x0_e, 3)):
        Xu = problem.synthetic_batch(x0, E.M, Bu, seed=77)
        dXu = torch.from_numpy(Xu).to(dev)
        dr = torch.empty((Bu, E.nres), dtype=torch.float64, device=dev)
        dj = torch.empty((Bu, E.V), dtype=torch.float64, device=dev)
        dupd = torch.full((Bu, E.total_nnz), float("nan"), dtype=torch.float64, device=dev)
        dref = torch.empty((Bu, E.total_nnz), dtype=torch.float64, device=dev)
        E.fill_full_device(Bu, dupd.data_ptr(), s)
        assert E.sync(s) == 0
        assert np.array_equal(dupd[Bu - 1].cpu().numpy()[~E.var_mask()], E.const_values()[~E.var_mask()])
        for scale in (1.0, 1.0 + 3e-7):      # two different batches through the same buffer
            dXs = dXu * scale
            E.eval_batch_device(Bu, dXs.data_ptr(), dr.data_ptr(), dj.data_ptr(), s)
            E.update_full_device(Bu, dj.data_ptr(), dupd.data_ptr(), s)
            E.expand_full_device(Bu, dj.data_ptr(), dref.data_ptr(), s)
            assert E.sync(s) == 0
            assert torch.equal(dupd, dref)
        # ... and in one call (evaluation + update; a launch that fits the Infinity Cache writes its compact values with ordinary
        # stores): the same residual rows, compact values and COO values, with and without residual rows
        dr2 = torch.full_like(dr, float("nan"))
        dj2 = torch.full_like(dj, float("nan"))
        dupd2 = torch.full((Bu, E.total_nnz), float("nan"), dtype=torch.float64, device=dev)
        E.fill_full_device(Bu, dupd2.data_ptr(), s)
        E.eval_full_device(Bu, dXs.data_ptr(), dr2.data_ptr(), dj2.data_ptr(), dupd2.data_ptr(), s)
        assert E.sync(s) == 0 and torch.equal(dr2, dr) and torch.equal(dj2, dj) and torch.equal(dupd2, dref)
        dupd2.fill_(float("nan"))
        E.fill_full_device(Bu, dupd2.data_ptr(), s)
        E.eval_full_device(Bu, dXu.data_ptr(), 0, dj2.data_ptr(), dupd2.data_ptr(), s)
        E.eval_batch_device(Bu, dXu.data_ptr(), 0, dj.data_ptr(), s)
        E.expand_full_device(Bu, dj.data_ptr(), dref.data_ptr(), s)
        assert E.sync(s) == 0 and torch.equal(dj2, dj) and torch.equal(dupd2, dref)


def test_self_signalling_launch_has_delivered_everything_when_it_says_so():
    """One-vector launches tell the host themselves when their results are in pinned host memory (the last workgroup stores the
    launch's sequence number after every workgroup's system-scope release) and the host returns on that word, not on the
    end-of-kernel signal.  Whatever the call hands back must already be final: snapshots taken the moment the call returns equal
    the buffers after a full synchronise, over many launches whose every value differs from the launch before -- the fused
    one-vector evaluation, the values-only and the values + derivatives callback with aero rows, and a NaN that must be reported."""
    from gelato_amd import Engine, con_dynamics, pack_x, problem
    pdict, unitdict, _c, xdict = problem.make_problem("mixed-6x64")
    E = Engine(con_dynamics.problem_arrays(pdict, unitdict))
    S = pdict["num_sections"]
    for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
        E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1) if pdict["params"][i]["reference_area"] != 0.0])
    pres, pvals = E.pinned_buffers()
    xb, xp = E.pinned_x()
    x0 = pack_x(xdict)
    prev_res, prev_vals = None, None
    for it in range(400):
        xb[it & 1][:] = x0 * (1.0 + 1e-9 * (it + 1))
        if it % 3 == 0:
            res, vals, rc = E.eval(xb[it & 1], out=pvals, res_out=pres)
            snap = (res.copy(), vals.copy())
            assert rc == 0 and E.sync() == 0
            assert np.array_equal(snap[0], pres) and np.array_equal(snap[1], pvals), it
            if prev_res is not None:
                assert not np.array_equal(snap[0], prev_res) and not np.array_equal(snap[1][E.var_mask()], prev_vals[E.var_mask()])
            prev_res, prev_vals = snap
        else:
            fr = E.eval_callback(xb[it & 1], it % 3 == 2, xptr=xp[it & 1])
            snap = {k: fr[k].copy() for k in ("res",)}
            snap_a = {k: v.copy() for k, v in fr["aero_con"].items()}
            snap_j = {k: v.copy() for k, v in (fr["aero_jac"] or {}).items()}
            snap_v = fr["vals"].copy() if fr["vals"] is not None else None
            assert fr["rc"] == 0 and E.sync() == 0
            assert np.array_equal(snap["res"], fr["res"]) and all(np.array_equal(snap_a[k], fr["aero_con"][k]) for k in snap_a), it
            assert all(np.array_equal(snap_j[k], fr["aero_jac"][k]) for k in snap_j), it
            assert snap_v is None or np.array_equal(snap_v, fr["vals"]), it
    xb[0][:] = x0
    xb[0][E.M + 7] = np.nan
    _, _, rc = E.eval(xb[0], out=pvals, res_out=pres)
    assert rc == 1
    fr = E.eval_callback(xb[0], True, xptr=xp[0])
    assert fr["rc"] == 1
    xb[0][:] = x0
    assert E.eval(xb[0], out=pvals, res_out=pres)[2] == 0


def test_self_signalling_calls_between_other_work_on_the_stream_and_on_two_handles():
    """The host returns from a one-vector call while the handle's stream still holds the kernel's tail: a batch launched right
    behind it through the device-pointer API, another one-vector call, and a second handle working in between all see what they
    should."""
    import torch
    from gelato_amd import Engine, con_dynamics, pack_x, problem
    pd1, ud1, _c, xd1 = problem.make_problem("mixed-6x64")
    pd2, ud2, _c, xd2 = problem.make_problem("example")
    E1, E2 = Engine(con_dynamics.problem_arrays(pd1, ud1)), Engine(con_dynamics.problem_arrays(pd2, ud2))
    x1, x2 = pack_x(xd1), pack_x(xd2)
    ref1 = E1.eval(x1)[:2]
    ref2 = E2.eval(x2)[:2]
    dev = torch.device("cuda:0")
    B = 64
    X = np.tile(x1, (B, 1))
    dX = torch.from_numpy(X).to(dev)
    dr = torch.empty((B, E1.nres), dtype=torch.float64, device=dev)
    dj = torch.empty((B, E1.V), dtype=torch.float64, device=dev)
    for it in range(60):
        r1, v1, rc1 = E1.eval(x1)
        E1.eval_batch_device(B, dX.data_ptr(), dr.data_ptr(), dj.data_ptr())      # the handle's own stream, right behind the call
        r2, v2, rc2 = E2.eval(x2)
        r1b, v1b, rc1b = E1.eval(x1 * (1.0 + 1e-9))
        assert rc1 == 0 and rc2 == 0 and rc1b == 0
        assert np.array_equal(r1, ref1[0]) and np.array_equal(v1, ref1[1]) and np.array_equal(r2, ref2[0]) and np.array_equal(v2, ref2[1])
        assert not np.array_equal(r1b, r1)
    assert E1.sync() == 0
    assert np.array_equal(dr[B - 1].cpu().numpy(), ref1[0]) and np.array_equal(E1.expand(dj[B - 1:].cpu().numpy())[0], ref1[1])


@pytest.mark.parametrize("nn", [(64,), (5, 33, 67), (70, 5), (128, 7), (2, 16, 4)])
def test_values_only_callback_splits_the_product_over_the_four_wavefronts(nn):
    """A values-only one-vector launch forms D.X as four row tiles, one per wavefront of the work item, from a state-row image staged
    once (phases up to 67 nodes) or from direct loads (longer phases): the residual rows are those of the evaluation with
    derivatives (where the lead wavefront multiplies alone) and of the residual-only batch launch, bit for bit."""
    import states
    prob, x = states.long_state(nn)
    E, _P = make_pair(prob)
    fr = E.eval_callback(x, False)
    res_v = fr["res"].copy()
    fr2 = E.eval_callback(x, True)
    assert fr["rc"] == 0 and fr2["rc"] == 0 and np.array_equal(res_v, fr2["res"])
    r1, rc = E.eval_residual(x)
    assert rc == 0 and np.array_equal(res_v, r1)
    rb, _, rc = E.eval_batch(np.tile(x, (8, 1)), want_jac=False)
    assert rc == 0 and np.array_equal(rb[5], res_v)


def test_placed_batch_buffers_hold_the_same_results():
    """gelato_amd.placement: candidate allocations of the resident x / res / jvar buffers, the fastest kept -- whichever it is, a launch
    on it writes the bits a launch on plain buffers writes, and the report names every candidate."""
    import torch
    from gelato_amd import problem
    from gelato_amd.placement import place_batch_buffers
    prob, x0, _ = named_problem("mixed-6x64")
    E, _P = make_pair(prob)
    B = 512
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    dX0 = torch.from_numpy(problem.synthetic_batch(x0, E.M, B, seed=5)).to(dev)
    dX, dres, djv, rep = place_batch_buffers(E, dX0, tries=3, launches=2, warm=1, stream=s)
    assert rep["tries"] == 3 and 0 <= rep["chosen"] < 3 and len(rep["candidates"]) == 3 and rep["all_launches"] == 9
    assert torch.equal(dX, dX0) and dX.data_ptr() != dX0.data_ptr()
    assert rep["candidates"][rep["chosen"]]["ms_per_launch"] == min(c["ms_per_launch"] for c in rep["candidates"])
    r = torch.empty_like(dres)
    j = torch.empty_like(djv)
    E.eval_batch_device(B, dX0.data_ptr(), r.data_ptr(), j.data_ptr(), s)
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    assert E.sync(s) == 0 and torch.equal(r, dres) and torch.equal(j, djv)
    dX2, dres2, djv2, rep2 = place_batch_buffers(E, dX0, want_jac=False, tries=2, launches=1, warm=0, stream=s)
    assert djv2 is None and rep2["tries"] == 2 and dres2.shape == (B, E.nres)


def test_nonfinite_input_sets_status():
    prob, x0, _ = named_problem("3x32")
    E, _ = make_pair(prob)
    x = x0.copy()
    x[E.M + 5] = np.nan
    res, rc = E.eval_residual(x)
    assert rc == 1 and np.isnan(res).any()
    res, rc = E.eval_residual(x0)      # the flag is cleared again
    assert rc == 0 and np.isfinite(res).all()


# --------------------------------------------------------------------------
# a20: generic column-batched forward difference
# --------------------------------------------------------------------------
def test_jac_fd_vs_golden_oracle_and_structured():
    oracle = _setup()
    g6 = load_golden("g6_example.npz")
    g7 = load_golden("g7_jacfd_example.npz")
    prob = problem_from_golden(g6)
    D, tau = D_tau_from_golden(g6, prob)
    E, P = make_pair(prob, D, tau, barC20=TW)
    x = g6["x"]
    vals, _ = E.eval_jacobian(x)
    Jd = E.jac_dicts(vals)
    off = {"mass": 0, "position": E.M, "velocity": 4 * E.M, "quaternion": 7 * E.M, "u": 11 * E.M,
           "t": 11 * E.M + 2 * E.N}
    for grp in oracle.GROUPS:
        J, rc = E.jac_fd(grp, x)
        assert rc == 0
        Xs = E.split_x(np.arange(E.nvars))
        for k, idx in Xs.items():
            ref = g7["%s_%s" % (grp, k)]
            d = np.abs(J[:, idx] - ref)
            assert np.all(d <= 1e-5 + 1e-6 * np.abs(ref)), (grp, k, d.max())
        d = np.abs(J - P.jac_fd(grp, x))
        assert d.max() <= 1e-5
        dense = np.zeros_like(J)
        for var, blk in Jd[grp].items():
            r, c, v = blk["coo"]
            dense[r, c + off[var]] += v
        assert np.max(np.abs(dense - J)) <= 1e-5         # structured COO == generic FD
        # the same quotients without the zeros: per-phase blocks, to the host and resident on the device -- the dense matrix's bits
        import torch
        blocks, rcb = E.jac_fd_blocks(grp, x)
        assert rcb == 0
        rebuilt = np.zeros_like(J)
        r_, c_, r0_, off_, cols_ = E.jac_fd_block_dims(grp)
        assert int(off_[-1]) == sum(b.size for _, _, b in blocks) and sum(int(v) for v in r_) == J.shape[0]
        for row0, cols, blk in blocks:
            assert len(set(cols.tolist())) == len(cols)
            rebuilt[row0:row0 + blk.shape[0], cols] = blk
        assert np.array_equal(rebuilt, J)
        dev = torch.device("cuda:0")
        s = torch.cuda.current_stream().cuda_stream
        dx_ = torch.from_numpy(x).to(dev)
        dJ = torch.full(J.shape, float("nan"), dtype=torch.float64, device=dev)
        E.jac_fd_device(grp, dx_.data_ptr(), dJ.data_ptr(), False, s)
        assert E.sync(s) == 0 and np.array_equal(dJ.cpu().numpy(), J)
        dB = torch.full((int(off_[-1]),), float("nan"), dtype=torch.float64, device=dev)
        E.jac_fd_device(grp, dx_.data_ptr(), dB.data_ptr(), True, s)
        assert E.sync(s) == 0
        hb = dB.cpu().numpy()
        assert all(np.array_equal(hb[int(off_[i]):int(off_[i + 1])].reshape(blk.shape), blk) for i, (_, _, blk) in enumerate(blocks))
        J2, _ = E.jac_fd(grp, x)                        # the host form after a device call (its kept residuals were overwritten)
        assert np.array_equal(J2, J)


# --------------------------------------------------------------------------
# boundary: the pyoptsparse callback surface
# --------------------------------------------------------------------------
def test_callback_surface_like_reference():
    oracle = _setup()
    from gelato_amd import con_dynamics, driver, jac_fd, problem
    g = load_golden("g6_example.npz")
    prob_ref = problem_from_golden(g)
    D, tau = D_tau_from_golden(g, prob_ref)

    class RefPS:  # the D / tau the caller put in pdict are inputs of the path
        def __init__(self, inner):
            self._i = inner
        def __getattr__(self, k):
            return getattr(self._i, k)
        def D(self, i):
            return D[i]
        def tau(self, i):
            return tau[i]

    pdict, unitdict, condition, xdict = problem.make_problem("example")
    pdict["ps_params"] = RefPS(pdict["ps_params"])
    pdict["barC20"] = TW
    xd = oracle.Problem(prob_ref).split_x(g["x"])
    xd = {k: v.copy() for k, v in xd.items()}
    keep = {k: v.copy() for k, v in xd.items()}
    objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
    funcs, fail = objfunc(xd)
    fs, fail2 = sens(xd, funcs)
    assert fail is False and fail2 is False
    assert all(np.array_equal(xd[k], keep[k]) for k in xd)            # xdict is never mutated
    assert funcs["obj"] == -xd["mass"][0]
    for grp, key in [("mass", "eqcon_dyn_mass"), ("pos", "eqcon_dyn_pos"), ("vel", "eqcon_dyn_vel"),
                     ("quat", "eqcon_dyn_quat")]:
        assert funcs[key].ndim == 1 and funcs[key].dtype == np.float64
        close(funcs[key], g["res_" + grp], what=key)
        assert list(fs[key].keys()) == driver.WRT[key]                 # wrt map, Trajectory_Optimization.py:361-364
        for var, blk in fs[key].items():
            r, c, v = blk["coo"]
            assert np.array_equal(r, g["jac_%s_%s_rows" % (grp, var)])
            assert np.array_equal(c, g["jac_%s_%s_cols" % (grp, var)])
            ref = g["jac_%s_%s_vals" % (grp, var)]
            assert np.all(np.abs(v - ref) <= 1e-5 + 1e-6 * np.abs(ref))
            assert blk["shape"] == tuple(g["jac_%s_%s_shape" % (grp, var)])
            assert len(set(zip(r.tolist(), c.tolist()))) == len(r)     # no duplicate (row, col)
    # second call with the same x is served from the cache, a moved x is re-evaluated
    f2, _ = objfunc(xd)
    assert np.array_equal(f2["eqcon_dyn_vel"], funcs["eqcon_dyn_vel"])
    xd2 = {k: v * (1 + 1e-9) for k, v in xd.items()}
    f3, _ = objfunc(xd2)
    assert not np.array_equal(f3["eqcon_dyn_vel"], funcs["eqcon_dyn_vel"])
    # jac_fd front end
    J = jac_fd.jac_fd(con_dynamics.equality_dynamics_mass, xd, pdict, unitdict, condition)
    assert set(J) == set(xd) and J["mass"].shape == (pdict["N"], pdict["M"])
    # a user's own Python function goes column by column like lib/jac_fd.py, without touching xd
    Ju = jac_fd.jac_fd(lambda xd_, *a: 2.0 * xd_["mass"][:3] + xd_["t"][-1], xd, pdict, unitdict, condition)
    assert set(Ju) == set(xd) and Ju["mass"].shape == (3, pdict["M"]) and np.allclose(Ju["mass"][:, :3], 2 * np.eye(3), atol=1e-6)
    assert np.allclose(Ju["t"][:, -1], 1.0, atol=1e-6) and not Ju["position"].any()
    assert all(np.array_equal(xd[k], keep[k]) for k in xd)
    stats = driver.mock_optimizer_loop(objfunc, sens, xd, iterations=3)
    assert stats["userObjCalls"] == 3 and stats["fails"] == 0
    # fresh value arrays by default (like the reference); shared ones on request
    a1 = con_dynamics.equality_jac_dynamics_velocity(xd, pdict, unitdict, condition)["velocity"]["coo"][2]
    a2 = con_dynamics.equality_jac_dynamics_velocity(xd, pdict, unitdict, condition)["velocity"]["coo"][2]
    assert a1 is not a2 and not np.shares_memory(a1, a2) and np.array_equal(a1, a2)
    pdict["gelato_amd_share_values"] = True
    b1 = con_dynamics.equality_jac_dynamics_velocity(xd, pdict, unitdict, condition)["velocity"]["coo"][2]
    b2 = con_dynamics.equality_jac_dynamics_velocity(xd, pdict, unitdict, condition)["velocity"]["coo"][2]
    assert np.shares_memory(b1, b2) and np.array_equal(b1, a1)


# --------------------------------------------------------------------------
# full-size properties (6 x 64, large batch): determinism, permutation, batch-position independence
# --------------------------------------------------------------------------
def test_full_size_batch_properties():
    import torch
    from gelato_amd import problem
    prob, x0, _ = named_problem("mixed-6x64")
    E, P = make_pair(prob)
    B = 2048
    X = problem.synthetic_batch(x0, E.M, 64)
    X = np.tile(X, (B // 64, 1))
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    s = torch.cuda.current_stream().cuda_stream

    def run(dx):
        r = torch.empty((dx.shape[0], E.nres), dtype=torch.float64, device=dev)
        jv = torch.empty((dx.shape[0], E.V), dtype=torch.float64, device=dev)
        E.eval_batch_device(dx.shape[0], dx.data_ptr(), r.data_ptr(), jv.data_ptr(), s)
        assert E.sync(s) == 0
        return r, jv

    r1, j1 = run(dX)
    r2, j2 = run(dX)
    assert torch.equal(r1, r2) and torch.equal(j1, j2)                  # deterministic
    assert torch.equal(r1[:64], r1[-64:]) and torch.equal(j1[:64], j1[-64:])   # independent of batch position
    perm = torch.randperm(B, device=dev)
    r3, j3 = run(dX[perm].contiguous())
    assert torch.equal(r3, r1[perm]) and torch.equal(j3, j1[perm])      # permutation equivariance
    # a sample of the big batch against the oracle
    ores, ovals = P.eval_batch(X[[0, 17, 63]])
    close(r1[[0, 17, 63]].cpu().numpy(), ores, what="full-size residual sample")
    full = E.expand(j1[[0, 17, 63]].cpu().numpy())
    vm = E.var_mask()
    d = np.abs(full - ovals)
    assert np.all(d[:, vm] <= 1e-5 + 1e-6 * np.abs(ovals[:, vm]))
    assert np.array_equal(full[:, ~vm], ovals[:, ~vm])


# --------------------------------------------------------------------------
# multi-GPU phase shards, exercised on one GPU: disjoint work-item ranges fill disjoint entries
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name,B", [("mixed-6x64", 3), ("example", 1), ("mixed-6x64", 300)])
def test_unit_shards_compose_to_the_full_evaluation(name, B):
    """BASELINE.json configs[3]: phases AND Jacobian columns dealt to GPUs.  unit = 4 * work item + part; every
    unit range writes only its own entries, and any disjoint cover sums to the unsharded result, bit for bit
    (B = 300 compares the unit form with the throughput form of the kernel)."""
    import torch
    from gelato_amd import parallel, problem
    prob, x0, _ = named_problem(name)
    E, _ = make_pair(prob)
    X = problem.synthetic_batch(x0, E.M, min(B, 16))
    X = np.tile(X, (B // len(X) + 1, 1))[:B]
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    ref_r = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    ref_j = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    E.eval_batch_device(B, dX.data_ptr(), ref_r.data_ptr(), ref_j.data_ptr(), s)
    assert E.sync(s) == 0
    costs = parallel.unit_costs(E)
    assert len(costs) == 4 * E.num_chunks()
    for world in (1, 3, 8, len(costs)):
        ranges = parallel.shard_chunks(costs, world)
        assert sum(c for _, c in ranges) == len(costs)
        if name == "mixed-6x64" and world == 8:
            assert all(c > 0 for _, c in ranges)              # 8 GPUs all get work on the 6-phase mesh
        tot_r = torch.zeros_like(ref_r)
        tot_j = torch.zeros_like(ref_j)
        for rank in range(world):
            r = torch.zeros_like(ref_r)
            jv = torch.zeros_like(ref_j)
            b, c = ranges[rank]
            E.eval_shard_units_device(B, dX.data_ptr(), r.data_ptr(), jv.data_ptr(), b, c, s)
            assert E.sync(s) == 0
            assert torch.all((tot_r == 0) | (r == 0)) and torch.all((tot_j == 0) | (jv == 0))   # disjoint owners
            tot_r += r
            tot_j += jv
        assert torch.equal(tot_r, ref_r) and torch.equal(tot_j, ref_j), (name, B, world)
    with pytest.raises(Exception):
        E.eval_shard_units_device(B, dX.data_ptr(), 0, ref_j.data_ptr(), 0, len(costs) + 1, s)


# --------------------------------------------------------------------------
# D.X on the matrix pipe (v_mfma_f64_16x16x4_f64) vs wavefront dot-products (VALU)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name,world,B", [("mixed-6x64", 8, 5), ("mixed-6x64", 3, 5), ("example", 2, 5), ("stress-12x128", 8, 2),
                                           ("mixed-6x64", 4, 260), ("ragged", 5, 3)])
def test_unit_shard_exchange_on_one_gpu(name, world, B):
    """parallel.UnitShards (what bench.py --mode phase-shard and the gloo tests drive) with the ENGINE writing: every rank's
    kernel writes the entries of its units STRAIGHT into its slice of a NaN-filled exchange buffer (no pack launch); the
    buffer with all slices written is what the in-place all-gather leaves on every rank; read through the plan's map -- and
    through the one-launch device gather -- it must equal the unsharded launch bit for bit (multi-chunk phases, ragged
    chunks, every phase type; B = 260 compares the unit form with the throughput form of the kernel)."""
    import torch
    from gelato_amd import Engine, parallel, problem
    if name == "ragged":
        from states import ragged_state
        prob, x0 = ragged_state()
    else:
        prob, x0, _ = named_problem(name)
    E = Engine(prob)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    X = problem.synthetic_batch(x0, E.M, min(B, 8), seed=3)
    X = np.tile(X, (B // len(X) + 1, 1))[:B]
    dX = torch.from_numpy(X).to(dev)
    ref_r = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    ref_j = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    E.eval_batch_device(B, dX.data_ptr(), ref_r.data_ptr(), ref_j.data_ptr(), s)
    assert E.sync(s) == 0
    sh = parallel.UnitShards(E, world, 0)
    out = sh.buffer(B, dev)
    out.fill_(float("nan"))
    for r in range(world):                                     # what the ranks do side by side, one after the other here
        E.eval_shard_packed_device(B, dX.data_ptr(), out.data_ptr(), r, s)
    assert E.sync(s) == 0
    res, jv = sh.gather(out)
    assert torch.equal(res, ref_r) and torch.equal(jv, ref_j)
    # every rank wrote its own slice only, and exactly as many entries as it owns (the padded tail stays NaN)
    for r in range(world):
        written = int((~torch.isnan(out[r, 0])).sum())
        assert written == sum(sh.counts[r]), (r, written, sh.counts[r])
    r2 = torch.full_like(ref_r, float("nan"))
    j2 = torch.full_like(ref_j, float("nan"))
    E.shard_unpack_device(B, out.data_ptr(), r2.data_ptr(), j2.data_ptr(), s)
    assert E.sync(s) == 0
    assert torch.equal(r2, ref_r) and torch.equal(j2, ref_j)
    # the slices are padded to the largest share; the cut balances COST, not bytes: never more than gathering whole buffers, and
    # within 1.6 x of the ideal (N-1)/N on the 6-phase mesh the mode is meant for
    assert sh.bytes_received_per_vector() <= 8 * (E.nres + E.V) * (world - 1) + 16 * world
    if name in ("mixed-6x64", "example"):
        assert sh.bytes_received_per_vector() <= 8 * (E.nres + E.V) * (world - 1) / world * 1.6 + 16
    with pytest.raises(Exception):
        E.eval_shard_packed_device(B, dX.data_ptr(), out.data_ptr(), world, s)
    # the plan is state of the handle.  A second UnitShards on the same Engine replaces it; the first object must not size
    # buffers or launch with its stale width any more, and the C-ABI refuses a launch for a plan the handle no longer holds
    from gelato_amd._lib import GelatoAmdError
    world2 = 3 if world != 3 else 5
    sh2 = parallel.UnitShards(E, world2, 0)
    with pytest.raises(RuntimeError, match="replaced"):
        sh.buffer(B, dev)
    with pytest.raises(RuntimeError, match="replaced"):
        sh.step(lambda o, r: None, out)
    with pytest.raises(GelatoAmdError):
        E.eval_shard_packed_device(B, dX.data_ptr(), out.data_ptr(), 0, s, plan=sh.plan)
    with pytest.raises(GelatoAmdError):
        E.shard_unpack_device(B, out.data_ptr(), r2.data_ptr(), j2.data_ptr(), s, plan=sh.plan)
    out2 = sh2.buffer(B, dev)
    for r in range(world2):
        E.eval_shard_packed_device(B, dX.data_ptr(), out2.data_ptr(), r, s, plan=sh2.plan)
    assert E.sync(s) == 0
    res2, jv2 = sh2.gather(out2)
    assert torch.equal(res2, ref_r) and torch.equal(jv2, ref_j)


@pytest.mark.parametrize("name", ["example", "mixed-6x64", "stress-12x128"])
def test_dx_mfma_and_valu_paths_agree(name):
    oracle = _setup()
    from gelato_amd import Engine
    prob, x, _ = named_problem(name)
    P = oracle.Problem(prob)
    D = [P.D(i) for i in range(P.S)]
    tau = [P.tau(i) for i in range(P.S)]
    Em = Engine(prob, D=D, tau=tau, flags=1)   # GEL_FLAG_DX_MFMA
    Ev = Engine(prob, D=D, tau=tau, flags=2)   # GEL_FLAG_DX_VALU
    rm, vm_, rc1 = Em.eval(x)
    rv, vv_, rc2 = Ev.eval(x)
    assert rc1 == 0 and rc2 == 0
    assert np.array_equal(vm_, vv_)                      # the Jacobian does not go through D.X
    bound = dx_roundoff_bound(Em, x)
    for grp, a, b in zip(oracle.GROUPS, Em.split_res(rm).values(), Ev.split_res(rv).values()):
        assert np.all(np.abs(a - b) <= 1e-15 + 2 * bound[grp]), grp
        close(a, P.residual(grp, x), atol=1e-12 + bound[grp], what="mfma residual " + grp)
        close(b, P.residual(grp, x), atol=1e-12 + bound[grp], what="valu residual " + grp)
    # residual-only launches (the jac_fd path) as well
    r1, _ = Em.eval_residual(x)
    r2, _ = Ev.eval_residual(x)
    assert np.array_equal(r1, rm) and np.array_equal(r2, rv)


def test_single_phase_long_tables_other_units():
    """S = 1, wind / CA tables longer than 32 rows (binary-search branch of the lookups), another dx and
    other units than the example's."""
    prob, _, _ = named_problem("example")
    rng = np.random.default_rng(11)
    prob = dict(prob)
    n = 37
    prob["num_nodes"] = np.array([n], dtype=np.int32)
    for k, v in [("thrust", 420000.0), ("massflow", 140.9), ("reference_area", 2.21), ("nozzle_area", 0.68)]:
        prob[k] = np.array([v])
    prob["engine_on"] = np.array([1], dtype=np.int32)
    prob["attitude_hold"] = np.array([0], dtype=np.int32)
    prob["units"] = np.array([20000.0, 6.4e6, 500.0, 2.0, 300.0])
    prob["dx"] = 1e-7
    alt = np.concatenate([[-1e8], np.linspace(0.0, 120e3, 40), [1e10]])
    prob["wind_table"] = np.column_stack([alt, 20 * np.sin(alt / 7e3), 15 * np.cos(alt / 9e3)])
    mach = np.concatenate([np.linspace(0.0, 6.0, 35), [100.0]])
    prob["ca_table"] = np.column_stack([mach, 0.3 + 0.35 * np.exp(-(mach - 1.1) ** 2)])
    E, P = make_pair(prob)
    assert E.S == 1 and E.N == n and E.M == n + 1
    th = 0.75 + np.linspace(0, 0.02, n + 1)
    R = (6378137.0 + np.linspace(200.0, 90e3, n + 1)) / 6.4e6
    pos = np.column_stack([R * np.cos(th) * 0.8, R * np.cos(th) * 0.6, R * np.sin(th)])
    vel = np.column_stack([np.linspace(0.3, 4.0, n + 1), np.linspace(0.6, 2.0, n + 1), np.linspace(0.1, 1.5, n + 1)])
    quat = rng.standard_normal((n + 1, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    x = np.concatenate([np.linspace(1.2, 0.4, n + 1), pos.ravel(), vel.ravel(), quat.ravel(),
                        0.5 * rng.standard_normal(2 * n), [0.1, 0.9]])
    check_against_oracle(E, P, x, "single-phase/long-tables")


def test_benchmark_batch_against_oracle_in_full():
    """Every decision vector of a bench-style batch (the exact generator bench.py uses, 512 vectors of the
    mixed 6 x 64 workload) against the oracle: residuals and all x-dependent Jacobian entries."""
    import os
    import torch
    from gelato_amd import problem
    prob, x0, _ = named_problem("mixed-6x64")
    E, P = make_pair(prob)
    B = 512
    X = problem.synthetic_batch(x0, E.M, B, seed=20260313)
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), s)
    assert E.sync(s) == 0
    res, jv = dres.cpu().numpy(), djv.cpu().numpy()
    vidx = E.var_index()
    bound = np.concatenate(list(dx_roundoff_bound(E, x0).values()))
    nthreads = max(1, min(64, len(os.sched_getaffinity(0))))
    worst_r = worst_j = 0.0
    for lo in range(0, B, 64):
        ores, ovals = P.eval_batch(X[lo:lo + 64], nthreads=nthreads)
        dr = np.abs(res[lo:lo + 64] - ores) - (1e-12 + 2 * bound + 1e-10 * np.abs(ores))
        ov = ovals[:, vidx]
        dj = np.abs(jv[lo:lo + 64] - ov) - 1e-6 * np.abs(ov)
        worst_r, worst_j = max(worst_r, dr.max()), max(worst_j, dj.max())
    assert worst_r <= 0.0, worst_r
    assert worst_j <= 1e-5, worst_j


# --------------------------------------------------------------------------
# the path's exp (the library's algorithm with scalar coefficient operands, csrc/gel_physics.h fexp) gives the library's bits
# --------------------------------------------------------------------------
@pytest.mark.gpu
def test_path_exp_bit_identical_to_the_library():
    _setup()
    from gelato_amd import dynamics
    rng = np.random.default_rng(78)
    x = np.concatenate([rng.uniform(-40.0, 40.0, 1 << 17), rng.uniform(-3.0, 0.0, 1 << 16), rng.uniform(-699.0, 699.0, 1 << 16),
                        [0.0, -0.0, 1e-300, -1e-300, 1.0, -1.0, 699.999, -699.999],
                        # wavefronts with a lane outside (-700, 700) take the library's: same bits trivially, results 0 / inf / nan included
                        [750.0, -750.0, 1e4, -1e4, np.inf, -np.inf, np.nan, 5.0]])
    out = dynamics.point_eval(14, x)
    assert np.array_equal(out[:, 0], out[:, 1], equal_nan=True)
    fin = np.isfinite(x) & (np.abs(x) < 700)
    assert np.max(np.abs(out[fin, 0] / np.exp(x[fin]) - 1.0)) < 4.5e-16      # and the library's is within 2 ulp of the host's


# --------------------------------------------------------------------------
# the path's guard-free fp64 sqrt / division (csrc/gel_physics.h) give the compiler's bits
# --------------------------------------------------------------------------
def test_guard_free_sqrt_and_division_bit_identical():
    _setup()
    from gelato_amd import dynamics
    rng = np.random.default_rng(77)
    n = 1 << 18
    a = 10.0 ** rng.uniform(-60, 60, n) * rng.uniform(1, 10, n)
    b = 10.0 ** rng.uniform(-60, 60, n) * rng.uniform(1, 10, n) * rng.choice([-1.0, 1.0], n)
    # the magnitudes the path actually feeds them: radii, speeds, O(1) ratios, table abscissae
    a[:4096] = rng.uniform(0.5, 2.0, 4096)
    a[4096:8192] = rng.uniform(6.3e6, 8.0e6, 4096) ** 2
    b[:4096] = rng.uniform(0.5, 2.0, 4096)
    out = dynamics.point_eval(7, np.stack([a, b], axis=1))
    assert np.array_equal(out[:, 0], out[:, 1])            # fsqrt == sqrt, every bit
    assert np.array_equal(out[:, 2], out[:, 3])            # fdiv  == a / b, every bit
    # and both are the correctly rounded results numpy computes on the host
    assert np.array_equal(out[:, 1], np.sqrt(a)) and np.array_equal(out[:, 3], a / b)


# --------------------------------------------------------------------------
# host batches larger than one staging slot go through the two-slot pipeline: same bits as the
# device-resident launch, ragged last sub-batch, out= reuse, non-finite status carried out
# --------------------------------------------------------------------------
def test_pipelined_host_batch_equals_resident_launch():
    import torch
    from gelato_amd import problem
    prob, x0, _ = named_problem("mixed-6x64")
    E, P = make_pair(prob)
    B = 333                                         # 16 MiB / (8 V) = 80 per slot -> 5 sub-batches, last one ragged
    assert B * 8 * E.V > (16 << 20)
    X = np.tile(problem.synthetic_batch(x0, E.M, 37), (B // 37 + 1, 1))[:B]
    X *= (1.0 + 1e-9 * np.arange(B))[:, None]
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    res, jv, rc = E.eval_batch(X)
    assert rc == 0
    assert np.array_equal(res, dres.cpu().numpy()) and np.array_equal(jv, djv.cpu().numpy())
    # a few rows against the oracle, across sub-batch borders
    rows = [0, 79, 80, 81, 159, 160, 332]
    ores, ovals = P.eval_batch(X[rows])
    close(res[rows], ores, what="pipelined residual")
    full = E.expand(jv[rows])
    vm = E.var_mask()
    assert np.all(np.abs(full - ovals)[:, vm] <= 1e-5 + 1e-6 * np.abs(ovals)[:, vm])
    # out= reuse, residual only and Jacobian only
    res2, jv2, _ = E.eval_batch(X, out=(res, jv))
    assert res2 is res and jv2 is jv
    r3, none, _ = E.eval_batch(X, want_jac=False)
    assert none is None and np.array_equal(r3, res)
    none, j4, _ = E.eval_batch(X, want_res=False)
    assert none is None and np.array_equal(j4, jv)
    with pytest.raises(ValueError):
        E.eval_batch(X, out=(res[:10], jv))
    # a NaN in the third sub-batch: status 1, the other vectors unaffected, next call clean
    Xb = X.copy()
    Xb[200, 5] = np.nan
    rb, jb, rc = E.eval_batch(Xb)
    assert rc == 1 and np.array_equal(rb[:200], res[:200]) and np.array_equal(jb[201:], jv[201:])
    assert E.eval_batch(X)[2] == 0


def test_all_atmosphere_layers_and_both_hemispheres():
    """One aerodynamic phase whose 64 nodes climb from 300 m below the ellipsoid to 700 km: every US-1976 layer
    (lapse, isothermal, the 91-110 km ellipse, the exponential above 120 km), the geopotential switch at 86 km,
    wind / CA clamps on both sides, southern and northern latitudes, Mach 0.03 ... 30, long flight times -- once with
    the dense-air nodes at moderate latitude (-54 .. +83 degrees along the climb), once with the climb STARTING at 80 degrees
    (sea-level air at 80 degrees: 3e-6 relative noise in the reference's own entries, covered by its derived allowance)."""
    import states
    for lo, hi in ((-0.95, 1.45), (1.40, -1.2)):
        prob, x = states.all_layers_state(lo, hi)
        E, P = make_pair(prob)
        check_against_oracle(E, P, x, "all-layers %.2f..%.2f" % (lo, hi), prob=prob)
        # the oracle's own altitudes really span the table (guards the construction)
        oracle = _setup()
        up = prob["units"][1]
        pos = x[E.M:4 * E.M].reshape(-1, 3)
        h = np.array([oracle.ecef2geodetic(*(pos[i] * up))[2] for i in (1, 64)])
        assert h[0] < 0.0 and h[1] > 6.0e5


@pytest.mark.parametrize("seed", range(6))
def test_random_problem_structures_values(seed):
    """Random phase structures (see tests/test_pattern_and_parallel_cpu.py) at random physical states: residuals
    and every Jacobian value against the oracle.  Small launches take the split latency form of the kernel, so
    this also pins that form on ragged, multi-chunk and mixed-type problems."""
    prob, _, _ = named_problem("example")
    rng = np.random.default_rng(2000 + seed)
    S = int(rng.integers(1, 9))
    prob = dict(prob)
    nn = rng.integers(2, 24, S)
    if seed % 2 == 0:
        nn[rng.integers(0, S)] = int(rng.integers(64, 100))
    prob["num_nodes"] = nn.astype(np.int32)
    on = rng.integers(0, 2, S)
    prob["engine_on"] = on.astype(np.int32)
    prob["thrust"] = np.where(on, rng.uniform(1e4, 5e5, S), 0.0)
    prob["massflow"] = np.where(on, rng.uniform(1.0, 150.0, S), 0.0)
    prob["reference_area"] = np.where(rng.integers(0, 2, S), rng.uniform(0.5, 3.0, S), 0.0)
    prob["nozzle_area"] = np.where(on, rng.uniform(0.0, 1.0, S), 0.0)
    prob["attitude_hold"] = rng.integers(0, 2, S).astype(np.int32)
    E, P = make_pair(prob)
    N, M = E.N, E.M
    up = prob["units"][1]
    lat = rng.uniform(-1.5, 1.5, M)                            # up to 86 deg: the reference's own FD noise is allowed for (fd_noise.py)
    lon = rng.uniform(-np.pi, np.pi, M)
    R = (6378137.0 - 21385.0 * np.sin(lat) ** 2 + rng.uniform(0.0, 150e3, M)) / up
    pos = np.column_stack([R * np.cos(lat) * np.cos(lon), R * np.cos(lat) * np.sin(lon), R * np.sin(lat)])
    vel = rng.standard_normal((M, 3)) * rng.uniform(0.05, 4.0, (M, 1))
    quat = rng.standard_normal((M, 4))
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    x = np.concatenate([0.2 + rng.random(M), pos.ravel(), vel.ravel(), quat.ravel(), rng.standard_normal(2 * N),
                        np.sort(rng.random(S + 1))])
    res1, vals1 = check_against_oracle(E, P, x, "random-%d" % seed, prob=prob)
    # the same vector inside a large batch goes through the throughput form of the kernel: same bits
    B = 200
    X = np.tile(x, (B, 1))
    res, jv, rc = E.eval_batch(X)
    assert rc == 0 and np.array_equal(res[B - 1], res1) and np.array_equal(E.expand(jv[B - 1]), vals1)


@pytest.mark.parametrize("nn", [(68,), (131,), (200, 5), (87, 129, 64), (96,), (128, 7), (69, 70, 71), (160,), (100, 95), (72, 64, 127)])
def test_long_phases_slab_staged_product(nn):
    """Phases of 68 nodes and more form D.X from 32-row slabs of state rows and a ring of A operands, both brought into LDS by
    LDS-DMA under hand-counted waits: 3 .. 7 slabs, ragged last slabs and ragged last chunks, a last state row that is multiplied
    on the vector unit (n a multiple of four) and lies in the last slab or in one of its own (n a multiple of 32), rows past the
    phase that are zeroed (n + 1 not a multiple of four) -- against the oracle and, the same vector inside a batch, against the
    split form."""
    import states
    prob, x = states.long_state(nn)
    E, P = make_pair(prob)
    res1, vals1 = check_against_oracle(E, P, x, "long-%s" % (nn,), prob=prob)
    for B in (300, 517):                                             # cooperative form (not the split latency form); ragged groups
        X = np.tile(x, (B, 1))
        X[B // 2] *= 1.0 + 1e-9
        res, jv, rc = E.eval_batch(X)
        assert rc == 0 and E.launch_info(B)[2] == 0
        assert np.array_equal(res[B - 1], res1) and np.array_equal(E.expand(jv[B - 1]), vals1)
        assert np.array_equal(res[0], res1) and not np.array_equal(res[B // 2], res1)
        r2, _, rc = E.eval_batch(X, want_jac=False)                  # the residual-only launch stages the same way
        assert rc == 0 and np.array_equal(r2, res)


@pytest.mark.parametrize("nn,hold", [((64, 63, 66), (1, 1, 0)), ((70, 128, 9), (1, 1, 1)), ((33, 67, 101), (0, 1, 1)), ((64,), (1,))])
def test_hold_phases_form_the_product_without_the_quaternion_columns(nn, hold):
    """Hold-type phases (lib/con_dynamics.py:521-522: quaternion rows q[1:] - q[0]) never read the quaternion columns of D.X; the
    cooperative form packs their vectors seven columns each and multiplies two column tiles instead of three (one-slab form and
    slab loop, last state row on the vector unit or in a k-step of its own).  The columns that are formed keep their bits: against
    the oracle, and the same vector inside a batch against the latency form (which multiplies all eleven columns)."""
    import states
    prob, x = states.long_state(nn)
    prob["attitude_hold"] = np.array(hold, dtype=np.int32)
    E, P = make_pair(prob)
    res1, vals1 = check_against_oracle(E, P, x, "hold-%s" % (nn,), prob=prob)
    for B in (260, 515):
        X = np.tile(x, (B, 1))
        X[B // 2] *= 1.0 + 1e-9
        res, jv, rc = E.eval_batch(X)
        assert rc == 0 and E.launch_info(B)[2] == 0                   # cooperative form, not the split latency form
        assert np.array_equal(res[B - 1], res1) and np.array_equal(E.expand(jv[B - 1]), vals1)
        assert np.array_equal(res[0], res1) and not np.array_equal(res[B // 2], res1)
        r2, _, rc = E.eval_batch(X, want_jac=False)
        assert rc == 0 and np.array_equal(r2, res)


@pytest.mark.parametrize("nn", [(70, 5), (69, 9, 4), (71, 3)])
def test_long_phase_rows_do_not_see_the_next_phase(nn):
    """The image of a long phase's last slab holds rows past the phase -- the next phase's first state rows, as they lie in x.  The
    last k-step multiplies them by columns of D that hold zeros, which is only harmless if they are numbers: they are zeroed in
    the image, so a NaN in the NEXT phase's first rows leaves this phase's residual rows and Jacobian values what they were."""
    import states
    prob, x = states.long_state(nn)
    E, P = make_pair(prob)
    B = 260
    X = np.tile(x, (B, 1))
    res0, jv0, rc = E.eval_batch(X)
    assert rc == 0 and E.launch_info(B)[2] == 0
    n0 = nn[0]
    Xn = X.copy()
    for k in range(3):                                   # the first three state rows of phase 1: mass, position, velocity, quaternion
        row = n0 + 1 + k
        Xn[:, row] = np.nan
        Xn[:, E.M + 3 * row:E.M + 3 * row + 3] = np.nan
        Xn[:, 4 * E.M + 3 * row:4 * E.M + 3 * row + 3] = np.inf
        Xn[:, 7 * E.M + 4 * row:7 * E.M + 4 * row + 4] = np.nan
    res1, jv1, rc1 = E.eval_batch(Xn)
    assert rc1 == 1                                      # phase 1 itself is not finite, and says so
    N = E.N
    rows0 = np.concatenate([np.arange(n0), N + np.arange(3 * n0), 4 * N + np.arange(3 * n0), 7 * N + np.arange(4 * n0)])
    assert np.array_equal(res1[:, rows0], res0[:, rows0]) and np.isfinite(res0[:, rows0]).all()
    full0, full1 = E.expand(jv0[:2]), E.expand(jv1[:2])
    pat = E.pattern()
    off = 0
    for b_, (r, c) in enumerate(pat):                   # every Jacobian value of phase 0's rows (block-local row numbering)
        sel = r < (E.block_shape[b_][0] // N) * n0
        assert np.array_equal(full1[:, off:off + len(r)][:, sel], full0[:, off:off + len(r)][:, sel]), b_
        off += len(r)


def test_jac_fd_of_a_long_phase_problem():
    """The phase-by-phase forward difference evaluates every phase as a one-phase problem: a long phase's sub-problem takes the
    slab loop on a local decision vector (its rows past the phase lie in that vector's own control block)."""
    import states
    prob, x = states.long_state((70, 5))
    E, P = make_pair(prob)
    for grp in ("mass", "vel"):
        J, rc = E.jac_fd(grp, x)
        assert rc == 0 and np.isfinite(J).all()
        assert np.abs(J - P.jac_fd(grp, x)).max() <= 1e-5 * max(1.0, np.abs(J).max())
        blocks, _ = E.jac_fd_blocks(grp, x)
        rebuilt = np.zeros_like(J)
        for row0, cols, blk in blocks:
            rebuilt[row0:row0 + blk.shape[0], cols] = blk
        assert np.array_equal(rebuilt, J)


def test_path_sincos_and_log_accuracy():
    """gel::fsincos (|x| <= 3 pi/4: latitudes, half Earth angles) and gel::flog_ratio (0.5 < x < 2: temperature ratios
    inside a layer) against numpy: at most 1 ulp resp. 2 ulp off the correctly rounded value, and not more often
    wrong than the library's own functions; outside those ranges they ARE the library's functions."""
    _setup()
    from gelato_amd import dynamics
    rng = np.random.default_rng(5)
    n = 1 << 19
    ang = np.concatenate([rng.uniform(-2.35, 2.35, n), np.pi / 2 + rng.uniform(-1e-3, 1e-3, n // 4) * rng.random(n // 4),
                          rng.uniform(-1e-3, 1e-3, n // 4), np.pi / 4 + rng.uniform(-1e-6, 1e-6, n // 4),
                          rng.uniform(-40.0, 40.0, n // 4)])
    rat = np.concatenate([rng.uniform(0.5, 2.0, n), 1.0 + rng.uniform(-1e-6, 1e-6, n // 2), rng.uniform(2.0, 40.0, n // 4),
                          rng.uniform(1e-3, 0.5, n // 4)])
    assert len(ang) == len(rat)
    out = dynamics.point_eval(8, np.stack([ang, rat], axis=1))

    def ulps(a, ref):
        return np.abs(a - ref) / np.spacing(np.abs(ref))
    us, uc = ulps(out[:, 0], np.sin(ang)), ulps(out[:, 1], np.cos(ang))
    ls, lc = ulps(out[:, 2], np.sin(ang)), ulps(out[:, 3], np.cos(ang))
    assert us.max() <= 1.0 and uc.max() <= 1.0
    assert np.mean(us > 0) <= np.mean(ls > 0) + 0.005 and np.mean(uc > 0) <= np.mean(lc > 0) + 0.005
    big = np.abs(ang) > 2.35619449019234492885
    assert np.array_equal(out[big, 0], out[big, 2]) and np.array_equal(out[big, 1], out[big, 3])
    ref = np.log(rat)
    inside = (rat > 0.5) & (rat < 2.0)
    ul = np.abs(out[:, 4] - ref) / np.spacing(np.maximum(np.abs(ref), 1e-300))
    assert ul[inside].max() <= 2.0
    assert np.array_equal(out[~inside, 4], out[~inside, 5])


def test_position_sweep_entries_do_not_depend_on_wavefront_neighbours():
    """A node within centimetres of an atmosphere-layer or wind-table break sends its WAVEFRONT through the recomputing fallback
    of the position sweeps.  A covered lane must write the same bits whether or not a neighbour did that -- in a two-vectors-per-
    wavefront launch the neighbour is another decision vector (round 3 took density, pressure and 1/a from the
    difference form but wind and the latitude pair from the recomputation inside the fallback).  Vector `xb` (nodes on the breaks)
    and vector `xc` (the same nodes 50 m higher: no fallback) in every pairing inside one launch."""
    import torch
    import states
    from gelato_amd import Engine
    prob, xb = states.layer_break_state(n_max=32)
    E = Engine(prob, flags=1)                     # matrix pipe: two vectors per wavefront (one 32-node phase)
    M = E.M
    xc = xb.copy()
    pos = xc[M:4 * M].reshape(-1, 3)
    pos *= (1.0 + 50.0 / (np.linalg.norm(pos, axis=1, keepdims=True) * prob["units"][1]))
    B = 1024                                      # the cooperative form (a handful of vectors would take the split form)
    assert E.launch_info(B)[4] == 1 and E.launch_info(B)[2] == 0
    pat = [xb, xc, xc, xb, xb, xb, xc, xc]        # (xb, xc), (xc, xb), (xb, xb), (xc, xc): the four pairings of one workgroup
    X = np.stack([pat[i % 8] for i in range(B)])
    res, jv, rc = E.eval_batch(X)
    assert rc == 0
    ib = [i for i in range(B) if pat[i % 8] is xb]
    ic = [i for i in range(B) if pat[i % 8] is xc]
    for idx in (ib, ic):
        assert all(np.array_equal(jv[idx[0]], jv[i]) for i in idx[1:]), "a vector's Jacobian bits depend on its wavefront neighbour"
        assert all(np.array_equal(res[idx[0]], res[i]) for i in idx[1:])
    assert not np.array_equal(jv[ib[0]], jv[ic[0]])
    # and the one-vector-per-wavefront form gives the same bits for both
    E1 = Engine(prob, flags=1 | 4)
    r1, j1, rc = E1.eval_batch(np.stack([xb, xc, xb, xc]))
    assert rc == 0 and np.array_equal(j1[0], jv[ib[0]]) and np.array_equal(j1[1], jv[ic[0]])


def test_packed_unit_shards_with_the_recomputing_form():
    """GEL_FLAG_FD_RECOMPUTE changes the slot layout of a node (separate t0 / tf and finite-difference quaternion slots: 62 / 50 / 42
    per node); the packed exchange layout follows it: every rank's slice through the plan's map equals the unsharded launch."""
    import torch
    from gelato_amd import Engine, parallel, problem
    prob, x0, _ = named_problem("mixed-6x64")
    E = Engine(prob, flags=8)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    B = 3
    dX = torch.from_numpy(problem.synthetic_batch(x0, E.M, B, seed=9)).to(dev)
    ref_r = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    ref_j = torch.empty((B, E.V), dtype=torch.float64, device=dev)
    E.eval_batch_device(B, dX.data_ptr(), ref_r.data_ptr(), ref_j.data_ptr(), s)
    assert E.sync(s) == 0
    sh = parallel.UnitShards(E, 8, 0)
    out = sh.buffer(B, dev)
    out.fill_(float("nan"))
    for r in range(8):
        E.eval_shard_packed_device(B, dX.data_ptr(), out.data_ptr(), r, s)
    assert E.sync(s) == 0
    res, jv = sh.gather(out)
    assert torch.equal(res, ref_r) and torch.equal(jv, ref_j)


# --------------------------------------------------------------------------
# COO-direct output of the one-vector latency path (gel_pinned_buffers, gel_eval_kernel.h "COO-DIRECT"): the kernel writes the
# all-x-dependent blocks of the full value vector straight into the handle's pinned array; the rest is scattered by the host
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["example", "3x32", "mixed-6x64", "dense-6x64", "stress-12x128", "ragged", "breaks", "polar"])
def test_coo_direct_one_vector_output_equals_the_compact_path(name):
    """Engine.eval into the handle's pinned buffers (zero-copy, COO-direct kernel output), into caller arrays, and through
    eval_callback, against the compact batch path + the gather map (an independent route: cooperative kernel forms, expand on the
    host): the same bits -- full chunks (LDS-tile stores), ragged chunks (per-lane stores), phases of several chunks, held and
    engine-off and NoAir phases, wavefronts with lanes the difference form does not cover (masked first pass + recomputing pass), and
    the same buffers re-used for another decision vector."""
    _setup()
    import states
    from gelato_amd import Engine, problem
    if name in ("ragged", "breaks", "polar"):
        prob, x0 = {"ragged": states.ragged_state, "breaks": lambda: states.with_coast_tail(states.layer_break_state),
                    "polar": lambda: states.with_coast_tail(states.polar_dense_state)}[name]()
    else:
        prob, x0, _ = named_problem(name)
    E = Engine(prob)
    pres, pvals = E.pinned_buffers()
    assert np.array_equal(pvals, E.const_values())            # the constants lie there before the first evaluation
    X = problem.synthetic_batch(x0, E.M, 3, seed=5)
    X[2] = x0 * (1.0 + 2e-7)
    res_b, jv_b, rc_b = E.eval_batch(np.concatenate([X, X, X])[:9])     # nine vectors: a cooperative form, not the latency form
    full_b = E.expand(jv_b)
    for b in (0, 1, 2, 0):
        r, v, rc = E.eval(X[b], out=pvals, res_out=pres)
        assert rc == 0 and r is pres and v is pvals
        assert np.array_equal(pres, res_b[b]) and np.array_equal(pvals, full_b[b]), (name, b)
        r2, v2, rc2 = E.eval(X[b])                              # caller arrays, constants filled
        assert rc2 == 0 and np.array_equal(r2, res_b[b]) and np.array_equal(v2, full_b[b])
        v3 = np.full(E.total_nnz, np.nan)
        v3[~E.var_mask()] = E.const_values()[~E.var_mask()]
        _, v3b, _ = E.eval(X[b], out=v3)                        # caller array that holds the constants already
        assert v3b is v3 and np.array_equal(v3, full_b[b])
        cb = E.eval_callback(X[b], True)
        assert cb["rc"] == 0 and np.array_equal(cb["res"], res_b[b]) and np.array_equal(cb["vals"], full_b[b])
        vj, rcj = E.eval_jacobian(X[b], out=pvals)
        assert rcj == 0 and np.array_equal(vj, full_b[b])
    # the decision vector itself in one of the handle's pinned buffers: read in place (the two buffers in turn)
    xb, xp = E.pinned_x()
    for k, b in enumerate((1, 2, 0)):
        np.copyto(xb[k & 1], X[b])
        cb = E.eval_callback(xb[k & 1], True, xptr=xp[k & 1])
        assert cb["rc"] == 0 and np.array_equal(cb["res"], res_b[b]) and np.array_equal(cb["vals"], full_b[b])
        r, v, rc = E.eval(xb[k & 1], out=pvals, res_out=pres)
        assert rc == 0 and np.array_equal(r, res_b[b]) and np.array_equal(v, full_b[b])
    # a non-finite input is reported through the pinned route as well, and the next call is clean
    xb = X[0].copy(); xb[E.M + 4] = np.nan
    _, _, rc = E.eval(xb, out=pvals, res_out=pres)
    assert rc == 1 and np.isnan(pres).any()
    _, _, rc = E.eval(X[1], out=pvals, res_out=pres)
    assert rc == 0 and np.array_equal(pvals, full_b[1]) and np.array_equal(pres, res_b[1])


def test_page_locked_caller_buffers_skip_the_staging_copies():
    """gel_eval_batch with page-locked caller buffers (torch pin_memory): H2D / D2H straight from / into them, sub-batch by sub-batch;
    the same bits as with pageable arrays (staged through the handle's slots); a mix of pinned and pageable buffers takes the staged path."""
    import torch
    from gelato_amd import Engine, problem
    prob, x0, _ = named_problem("mixed-6x64")
    E = Engine(prob)
    B = 300                       # > one staging slot (16 MB / 136 KB = 117 vectors): three sub-batches, the last one ragged
    X = problem.synthetic_batch(x0, E.M, B, seed=11)
    res, jv, rc = E.eval_batch(X)
    assert rc == 0
    xp = torch.from_numpy(X).pin_memory()
    rp = torch.full((B, E.nres), float("nan"), dtype=torch.float64).pin_memory()
    jp = torch.full((B, E.V), float("nan"), dtype=torch.float64).pin_memory()
    r2, j2, rc2 = E.eval_batch(xp.numpy(), out=(rp.numpy(), jp.numpy()))
    assert rc2 == 0 and np.array_equal(r2, res) and np.array_equal(j2, jv)
    jpage = np.full((B, E.V), np.nan)
    r3, j3, rc3 = E.eval_batch(xp.numpy(), out=(rp.numpy(), jpage))       # one pageable buffer: staged
    assert rc3 == 0 and np.array_equal(j3, jv) and np.array_equal(r3, res)
    Xn = xp.numpy().copy(); Xn[250, 7] = np.nan
    xp2 = torch.from_numpy(Xn).pin_memory()
    r4, j4, rc4 = E.eval_batch(xp2.numpy(), out=(rp.numpy(), jp.numpy()))
    assert rc4 == 1 and np.array_equal(r4[:250], res[:250]) and np.array_equal(j4[251:], jv[251:])
