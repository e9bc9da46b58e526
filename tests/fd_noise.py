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
