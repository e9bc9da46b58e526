"""Derived bound of the rounding noise in the REFERENCE's finite-difference Jacobian of the velocity defect (what two fp64 runs
of src/pybind_dynamics.cpp:42-68, differenced and divided by dx = 1e-8, can be off from the exact quotient), per node of an
aerodynamic phase.  Test infrastructure: it calls the oracle.

A forward difference divides the rounding error of f(x') - f(x) by dx, i.e. multiplies it by 1e8.  Two sources:

 (a) every operation of the chain whose input moved with the perturbed variable rounds independently in the two runs: the
     thrust, aerodynamic and gravity terms each carry a few eps of THEIR OWN magnitude (not of their sum):
         C_CHAIN * eps * (|T d / m| + |F_aero / m| + |g|) / unit_vel;
 (b) position sweeps only -- the altitude  alt = p / cos(lat) - N  (src/Earth.cpp:58-59) cancels two numbers of 6.4e6 m:
     p / cos(lat) carries eps * (p / cos lat) * (1 + |lat tan lat| * C_LAT) (p, the division, and cos(lat) with lat from atan2:
     d cos = sin(lat) * eps * |lat| per ulp of lat), N carries eps * N; both runs round independently, and the atmosphere turns
     the altitude error into  |d f / d alt| * d_alt.

The entry's bound is (a + 2 * b) / dx * (tf - to) * unit_t / 2.  tests/test_exact_fd.py checks the oracle against
exact-arithmetic quotients (tests/golden/g15_exact_fd.npz) within this bound, and the engine's exact-difference form within
(a) alone with a smaller constant."""
import numpy as np

EPS = np.finfo(np.float64).eps
C_CHAIN = 64.0       # operations between the perturbed input and the result, incl. pow() with |exponent| <= 35
C_CHAIN_ENGINE = 32.0
C_LAT = 2.0


def velocity_noise_terms(orc, prob, x, barC20=None):
    """-> list over phases: None (no aerodynamics) or dict of per-node arrays: `chain` = sum of the magnitudes of the thrust,
    aerodynamic and gravity terms / unit_vel, `dfdalt` = max-norm of d(acc/unit_vel)/d(altitude) per metre, `dalt` = altitude
    rounding error of one run (m), `scale` = (tf - to) * unit_t / 2 / dx."""
    bc = orc.BARC20_CPP if barC20 is None else barC20
    nn = [int(v) for v in prob["num_nodes"]]
    S, N = len(nn), sum(nn)
    M = N + S
    units = np.asarray(prob["units"], dtype=np.float64)
    um, up, uv, uu, ut = units
    xm, xr, xv, xq = x[:M], x[M:4 * M].reshape(-1, 3), x[4 * M:7 * M].reshape(-1, 3), x[7 * M:11 * M].reshape(-1, 4)
    xt = x[11 * M + 2 * N:]
    out = []
    ua = 0
    for i, n in enumerate(nn):
        xa = ua + i
        if prob["reference_area"][i] == 0.0:
            out.append(None)
            ua += n
            continue
        sl = slice(xa + 1, xa + n + 1)
        to, tf = xt[i], xt[i + 1]
        tau = np.asarray(prob["tau"][i])
        tn = tau * (tf - to) / 2 + (tf + to) / 2
        param = np.array([prob["thrust"][i], prob["massflow"][i], prob["reference_area"][i], 0.0, prob["nozzle_area"][i]])
        f_air = orc.dynamics_velocity(xm[sl], xr[sl], xv[sl], xq[sl], tn, param, prob["wind_table"], prob["ca_table"], units[:3], bc)
        p0 = param.copy(); p0[0] = 0.0; p0[4] = 0.0
        f_drag = orc.dynamics_velocity(xm[sl], xr[sl], xv[sl], xq[sl], tn, p0, prob["wind_table"], prob["ca_table"], units[:3], bc)  # F/m + g
        p1 = param.copy(); p1[2] = 0.0
        f_thr = orc.dynamics_velocity_NoAir(xm[sl], xr[sl], xq[sl], p1, units[:3], bc)     # T_vac d / m + g
        pg = p1.copy(); pg[0] = 0.0
        f_g = orc.dynamics_velocity_NoAir(xm[sl], xr[sl], xq[sl], pg, units[:3], bc)       # g
        chain = (np.abs(f_drag - f_g) + np.abs(f_thr - f_g) + np.abs(f_g) + np.abs(f_air)).max(axis=1)
        # d f / d altitude by a 10 m radial step (its own rounding error is 1e-16 * 6.4e6 / 10: irrelevant)
        r = xr[sl] * up
        rn = np.linalg.norm(r, axis=1, keepdims=True)
        f_up = orc.dynamics_velocity(xm[sl], (r * (1.0 + 10.0 / rn)) / up, xv[sl], xq[sl], tn, param, prob["wind_table"], prob["ca_table"], units[:3], bc)
        dfdalt = np.abs(f_up - f_air).max(axis=1) / 10.0
        geo = np.array([orc.ecef2geodetic(*row) for row in r])      # deg, deg, m
        lat = np.deg2rad(geo[:, 0])
        q = np.hypot(r[:, 0], r[:, 1]) / np.maximum(np.abs(np.cos(lat)), 1e-300)
        dalt = EPS * (q * (1.0 + C_LAT * np.abs(lat * np.tan(lat))) + 6.4e6)
        # nodes whose perturbed point may sit in another atmosphere layer / piece of the wind table / geopotential branch than the
        # node itself (within one position step dx * unit_position of a break): the engine recomputes those like the reference
        alt = geo[:, 2]
        h = np.where(alt < 86000.0, 6356766.0 * alt / (6356766.0 + alt), alt)
        breaks = np.concatenate([[11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0, 84852.0, 86000.0, 91000.0, 110000.0, 120000.0],
                                 np.asarray(prob["wind_table"])[:, 0]])
        near = np.abs(h[:, None] - breaks[None, :]).min(axis=1) <= 1.5 * abs(prob["dx"] * up)
        out.append({"chain": chain, "dfdalt": dfdalt, "dalt": dalt, "scale": (tf - to) * ut / 2 / prob["dx"], "lat": lat, "alt": alt,
                    "near_break": near})
        ua += n
    return out


def reference_bound(terms):
    """per node: what the reference's vel/position entries may be off the exact quotient"""
    return (C_CHAIN * EPS * terms["chain"] + 2.0 * terms["dfdalt"] * terms["dalt"]) * np.abs(terms["scale"])


def reference_bound_other(terms):
    """... its vel/mass, vel/velocity, vel/quaternion, vel/t entries (no altitude term)"""
    return C_CHAIN * EPS * terms["chain"] * np.abs(terms["scale"])


def engine_bound(terms):
    """per node: what the engine's exact-difference entries may be off the exact quotient (no altitude term) -- except at nodes
    within a step of a break of the atmosphere / wind tables, where the engine recomputes like the reference"""
    return np.where(terms["near_break"], reference_bound(terms), C_CHAIN_ENGINE * EPS * terms["chain"] * np.abs(terms["scale"]))


# ---------------------------------------------------------------------------------------------------------------------------
# Aero path constraints (lib/con_aero.py:311-471): f = alpha / limit, q / limit or q alpha / limit, gradient = -(f_p - f_c)/dx.
#
# One evaluation of the chain (src/wrapper_utils.hpp:89-111,163-175) is off the exact value by
#   angle of attack   e_alpha = eps (C_ACOS / sin(alpha) + C_DIR A)
#       alpha = acos(c), c = v_air . dir / (|v_air| |dir|) ~ 1: the few roundings of c (absolute C_ACOS eps) move alpha by
#       eps / sin(alpha); the direction of v_air = R(theta) R(-theta)(v - omega x r) - w_eci carries the roundings of its terms,
#       whose magnitudes exceed |v_air| by A = (|v| + omega |r| + |w|) / |v_air|;
#   dynamic pressure  e_q = eps q (C_Q (1 + A) [+ C_RHO])    (|v_air|^2; the density's pow / exp chain: position sweeps only --
#       the other sweeps look the atmosphere up at the same altitude bits in both evaluations and its rounding cancels);
#   position sweeps   + |d alpha / d alt| d_alt, |d q / d alt| d_alt with the altitude's rounding d_alt of the velocity bound above.
# An entry differences two evaluations: bound = 2 e_f / dx.  The REFERENCE adds a drift: it perturbs in place (`+= dx`, `-= dx`,
# con_aero.py:335-360), which can leave a component one ulp off for every later sweep: |g_i| ulp(x_i) / dx per earlier component.
# ---------------------------------------------------------------------------------------------------------------------------
C_ACOS = 8.0
C_DIR = 8.0
C_Q = 4.0
C_RHO = 64.0
OMEGA_E = 7.2921151467e-5


def aero_noise_terms(orc, prob, x, spec):
    """spec: rows of (phase, range_all, ...) -> per constrained node (rows in the constraint's order): alpha, q, sin(alpha), the
    cancellation factor A, d_alt, |d alpha / d alt|, |d q / d alt|, and the node's normalised values (for the drift term)"""
    import ctypes as C
    L = orc.lib()
    dp = C.POINTER(C.c_double)
    L.orc_dynamic_pressure_pa.restype = C.c_double
    L.orc_angle_of_attack_all_rad.restype = C.c_double
    nn = [int(v) for v in prob["num_nodes"]]
    S, N = len(nn), sum(nn)
    M = N + S
    up, uv, ut = (float(prob["units"][k]) for k in (1, 2, 4))
    xr, xv, xq = x[M:4 * M].reshape(-1, 3), x[4 * M:7 * M].reshape(-1, 3), x[7 * M:11 * M].reshape(-1, 4)
    xt = x[11 * M + 2 * N:]
    wind = np.ascontiguousarray(prob["wind_table"], dtype=np.float64)
    Kw = len(wind)
    wmax = float(np.abs(wind[:, 1:]).max())

    def point(pos, vel, quat, t):
        a = L.orc_angle_of_attack_all_rad(pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), quat.ctypes.data_as(dp), C.c_double(t),
                                          wind.ctypes.data_as(dp), C.c_int(Kw))
        q = L.orc_dynamic_pressure_pa(pos.ctypes.data_as(dp), vel.ctypes.data_as(dp), C.c_double(t), wind.ctypes.data_as(dp), C.c_int(Kw))
        return a, q

    rows = []
    for sp in spec:
        ph, all_nodes = int(sp[0]), int(sp[1])
        xa = sum(nn[:ph]) + ph
        to, tf = float(xt[ph]), float(xt[ph + 1])
        tau = np.asarray(prob["tau"][ph], dtype=np.float64)
        for k in range(nn[ph] + 1 if all_nodes else 1):
            t = (to if k == 0 else tau[k - 1] * (tf - to) / 2 + (tf + to) / 2) * ut
            pos, vel, quat = xr[xa + k] * up, xv[xa + k] * uv, np.ascontiguousarray(xq[xa + k])
            a, q = point(pos, vel, quat, t)
            rn = np.linalg.norm(pos)
            a2, q2 = point(pos * (1.0 + 10.0 / rn), vel, quat, t)
            lat_deg, _, alt = orc.ecef2geodetic(*pos)
            lat = np.deg2rad(lat_deg)
            nv = np.linalg.norm(vel - np.cross([0.0, 0.0, OMEGA_E], pos)) - wmax      # a lower bound of |v_air|
            A = (np.linalg.norm(vel) + OMEGA_E * rn + wmax) / nv if nv > 0.0 else np.inf
            qq = np.hypot(pos[0], pos[1]) / max(abs(np.cos(lat)), 1e-300)
            dalt = EPS * (qq * (1.0 + C_LAT * abs(lat * np.tan(lat))) + 6.4e6)
            rows.append((a, q, np.sin(a), A, dalt, abs(a2 - a) / 10.0, abs(q2 - q) / 10.0,
                         np.concatenate([np.abs(xr[xa + k]), np.abs(xv[xa + k]), np.abs(xq[xa + k])]), lat_deg))
    keys = ("alpha", "q", "sin", "A", "dalt", "dalpha_dalt", "dq_dalt", "xabs", "lat_deg")
    return {k: np.array([r[i] for r in rows]) for i, k in enumerate(keys)}


def aero_bound(terms, kind, limit, dx, position):
    """per row: how far one fp64 implementation's gradient entry of `kind` (position columns / the other columns) may be from the
    exact quotient.  limit: per row (units[3] of con_aero.py)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        e_alpha = EPS * (C_ACOS / terms["sin"] + C_DIR * terms["A"])
        e_q = EPS * terms["q"] * (C_Q * (1.0 + terms["A"]) + (C_RHO if position else 0.0))
        e_alpha = np.where(terms["alpha"] > 0.0, e_alpha, np.inf)     # c_alpha > 1 is clamped to 0: not differentiable there
        e_q = np.where(np.isnan(e_q), 0.0, e_q)                       # no air: q = 0 in every evaluation (0 * inf); q > 0 where the
                                                                      # air-relative speed may vanish (lift-off): unbounded, not checked
    if position:
        e_alpha = e_alpha + terms["dalpha_dalt"] * terms["dalt"]
        e_q = e_q + terms["dq_dalt"] * terms["dalt"]
    def times(a, e):     # a * e with 0 * inf = 0: a factor that is exactly zero in every evaluation (no air; the clamped angle) carries no noise
        with np.errstate(invalid="ignore"):
            return np.where(a == 0.0, 0.0, a * e)
    e_f = {"alpha": e_alpha, "q": e_q, "qalpha": times(terms["q"], e_alpha) + times(terms["alpha"], e_q)}[kind]
    return 2.0 * e_f / dx / limit


def aero_drift(terms, grad_abs, dx):
    """per row: the reference's in-place drift.  grad_abs [R, 10]: magnitudes of the row's position, velocity and quaternion entries"""
    return (grad_abs * EPS * (terms["xabs"] + dx)).sum(axis=1) / dx


def aero_coo_bounds(orc, prob, x, kind, spec, drift_of=None):
    """{var: bound of every gradient entry, in the reference's emission order (per spec row, component-major: con_aero.py:437-463)}
    for ONE implementation against the exact quotient.  drift_of: {var: values in the same order} of a reference-style
    implementation -- adds the drift of its in-place perturbations, estimated from its own entries."""
    terms = aero_noise_terms(orc, prob, x, spec)
    nn = [int(v) for v in prob["num_nodes"]]
    blocks, lim, r0 = [], [], 0
    for sp in spec:
        nk = nn[int(sp[0])] + 1 if int(sp[1]) else 1
        blocks.append((r0, nk))
        lim += [float(sp[2])] * nk
        r0 += nk
    lim = np.array(lim)
    dx = float(prob["dx"])
    width = {"position": 3, "velocity": 3, "quaternion": 0 if kind == "q" else 4, "t": 2}
    drift = 0.0
    if drift_of is not None:
        g = np.zeros((r0, 10))
        for var, c0 in (("position", 0), ("velocity", 3), ("quaternion", 6)):
            w, off = width[var], 0
            for b0, nk in blocks:
                if w:
                    g[b0:b0 + nk, c0:c0 + w] = np.abs(np.asarray(drift_of[var])[off:off + w * nk]).reshape(w, nk).T
                off += w * nk
        drift = aero_drift(terms, g, dx)
    out = {}
    for var, w in width.items():
        b = aero_bound(terms, kind, lim, dx, position=(var == "position")) + drift
        out[var] = np.concatenate([np.repeat(b[b0:b0 + nk][None, :], w, axis=0).ravel() for b0, nk in blocks]) if w and blocks else np.zeros(0)
    return out


def aero_coo_rows(prob, kind, spec):
    """{var: the constraint row of every gradient entry, in the reference's emission order} (as aero_coo_bounds lays its bounds out)"""
    nn = [int(v) for v in prob["num_nodes"]]
    blocks, r0 = [], 0
    for sp in spec:
        nk = nn[int(sp[0])] + 1 if int(sp[1]) else 1
        blocks.append((r0, nk))
        r0 += nk
    width = {"position": 3, "velocity": 3, "quaternion": 0 if kind == "q" else 4, "t": 2}
    return {var: (np.concatenate([np.tile(np.arange(b0, b0 + nk), w) for b0, nk in blocks]) if w and blocks else np.zeros(0, dtype=int))
            for var, w in width.items()}


def aero_row_limits(prob, spec):
    """per constraint row: the limit (units[3] of lib/con_aero.py) of its spec"""
    nn = [int(v) for v in prob["num_nodes"]]
    return np.concatenate([[float(sp[2])] * (nn[int(sp[0])] + 1 if int(sp[1]) else 1) for sp in spec]) if len(spec) else np.zeros(0)


def aero_stated_tolerance(bound, ref):
    """THE stated tolerance of SURVEY row f-1, per gradient entry, two fp64 implementations of the reference's
    forward difference against each other:

        |e_a - e_b|  <=  1e-5 + 1e-6 |e_ref|  +  2 K / (dx limit)

    flat part: SURVEY 8(c)'s FD tolerance; K: what ONE evaluation of the row's function is off (aero_bound above, closed form
    in the node's alpha, q, A = (|v| + omega |r| + |w|) / |v_air| and the altitude rounding; constants C_ACOS = C_DIR = 8,
    C_Q = 4, C_RHO = 64, C_LAT = 2), doubled for the two evaluations of a difference, doubled again for two implementations.
    `bound` = aero_coo_bounds(...)[var] already holds 2 K / (dx limit) (+ the reference's in-place drift)."""
    return 1e-5 + 1e-6 * np.abs(ref) + 2.0 * bound


def aero_first_order_truncation(terms, rows, limit, kind, t, dx):
    """What the engine's entries would move by if its alpha-difference (gel_kernels.hip aero_dalpha: t = t0 (1 - x + 2 x^2 - x^3),
    x = cot(alpha_c) t0 / 2) were cut after its first term, t = t0, given the true differences t = alpha_p - alpha_c of every entry
    (from the oracle's alpha rows of the same nodes): the alpha part of an entry -- -t / (dx L) for the alpha kind, -q t / (dx L) for
    the q-alpha kind (q_p = q_c to first order) -- grows by the factor x.  q kind: zero."""
    if kind == "q":
        return np.zeros_like(t)
    a, q = terms["alpha"][rows], terms["q"][rows]
    with np.errstate(all="ignore"):
        x = np.where(np.sin(a) > 0.0, np.cos(a) / (2.0 * np.sin(a)) * t, 0.0)
        d = x * (-t / (dx * limit)) * (1.0 if kind == "alpha" else q)
    return np.where(np.isfinite(d), d, 0.0)
