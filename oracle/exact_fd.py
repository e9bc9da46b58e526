"""Exact-arithmetic evaluation of the reference's velocity RHS and of its forward-difference quotients (TEST INFRASTRUCTURE,
like everything under oracle/; needs mpmath, used by tests/golden/make_exact_fd.py in the build container).

What a finite-difference Jacobian entry of the reference IS, separated from how two fp64 runs of the chain happen to round:
the reference perturbs a normalised variable in fp64 (`x += dx`, lib/con_dynamics.py:362-480), scales it in fp64
(`* unit`, src/pybind_dynamics.cpp:33-35: a rounded product) and then runs src/pybind_dynamics.cpp:42-68; the entry is
-(f(x') - f(x))/dx * (tf - to) * unit_t / 2.  Here f is evaluated on exactly those fp64 inputs in 40-digit arithmetic
(mpmath), formula by formula as the C++ does (file:line cited per step), so the quotient carries the reference's truncation
error (it is the same finite step) but none of its rounding noise.  A correct fp64 implementation of ANY form -- the
reference's recomputation or the engine's exact-difference form -- lies within its own derivable rounding bound of these values;
tests/test_exact_fd.py asserts both bounds.
"""
import numpy as np
from mpmath import mp, mpf, sqrt, atan2, sin, cos, exp, power

mp.dps = 40

# src/Earth.cpp:41-47
MU = mpf("3.986004418e14")
OMEGA = mpf("7.2921151467e-5")
RA = mpf(6378137)
ONE_F = mpf("298.257223563")
RB = RA * (1 - 1 / ONE_F)
E2 = (RA * RA - RB * RB) / RA / RA
EP2 = (RA * RA - RB * RB) / RB / RB
# src/Air.cpp:28-45
RSTAR, G0, R0 = mpf("8314.32"), mpf("9.80665"), mpf(6356766)
HB = [mpf(v) for v in ("0", "11000", "20000", "32000", "47000", "51000", "71000", "86000", "91000", "110000", "120000")]
LMB = [mpf(v) for v in ("-0.0065", "0", "0.001", "0.0028", "0", "-0.0028", "-0.002", "0", "0.0025", "0.012", "0.012")]
TMB = [mpf(v) for v in ("288.15", "216.65", "216.65", "228.65", "270.65", "270.65", "214.65", "186.8673", "186.8673", "240.0", "360.0")]
PB = [mpf(v) for v in ("101325.0", "22632.0", "5474.9", "868.02", "110.91", "66.939", "3.9564", "0.37338", "0.15381", "7.1042e-3", "2.5382e-3")]
MB = [mpf(v) for v in ("28.9644", "28.9644", "28.9644", "28.9644", "28.9644", "28.9644", "28.9644", "28.9522", "28.89", "27.27", "26.20")]


def f64(v):
    """an fp64 number as an exact mpf"""
    return mpf(float(v))


def geodetic(x, y, z):
    """src/Earth.cpp:49-61 -> (lat, lon, alt), radians"""
    p = sqrt(x * x + y * y)
    th = atan2(z * RA, p * RB)
    lat = atan2(z + EP2 * RB * sin(th) ** 3, p - E2 * RA * cos(th) ** 3)
    lon = atan2(y, x)
    N = RA / sqrt(1 - E2 * sin(lat) ** 2)
    return lat, lon, p / cos(lat) - N


def atmosphere(h):
    """src/Air.cpp:56-111 at geopotential altitude h -> (T, P, rho, a)"""
    k = 0
    for i in range(11):
        if h >= HB[i]:
            k = i
    Hb, Lmb, Tmb, Pb, R = HB[k], LMB[k], TMB[k], PB[k], RSTAR / MB[k]
    if h <= 91000:
        T = Tmb + Lmb * (h - Hb)
    elif h <= 110000:
        a_ = mpf("-19942.9")
        T = mpf("263.1905") + mpf("-76.3232") * sqrt(1 - (h - 91000) * (h - 91000) / a_ / a_)
    elif h <= 120000:
        T = Tmb + Lmb * (h - Hb)
    else:
        xi = (h - Hb) * (R0 + Hb) / (R0 + h)
        T = 1000 - (1000 - Tmb) * exp(mpf("-0.01875e-3") * xi)
    if abs(Lmb) > mpf("1e-6"):
        P = Pb * power((Tmb + Lmb * (h - Hb)) / Tmb, -G0 / Lmb / R)
    else:
        P = Pb * exp(G0 / R * (Hb - h) / Tmb)
    return T, P, P / R / T, sqrt(mpf("1.4") * R * T)


def interp(x, xp, yp):
    """src/wrapper_utils.hpp:51-80 with np.interp's value at x == xp[0]"""
    n = len(xp)
    if x <= xp[0]:
        return yp[0]
    if x > xp[n - 1]:
        return yp[n - 1]
    idx = 0
    while xp[idx + 1] < x:
        idx += 1
    return yp[idx] + (x - xp[idx]) / (xp[idx + 1] - xp[idx]) * (yp[idx + 1] - yp[idx])


def quatmult(q, p):  # src/wrapper_coordinate.hpp:50-57
    return [q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3],
            q[0] * p[1] + q[1] * p[0] + q[2] * p[3] - q[3] * p[2],
            q[0] * p[2] - q[1] * p[3] + q[2] * p[0] + q[3] * p[1],
            q[0] * p[3] + q[1] * p[2] - q[2] * p[1] + q[3] * p[0]]


def conj(q):
    return [q[0], -q[1], -q[2], -q[3]]


def quatrot(q, v):  # :70-78
    return quatmult(conj(q), quatmult([mpf(0)] + list(v), q))[1:]


def gravity(r, barC20):
    """src/gravity.cpp:11-57"""
    x, y, z = r
    rn = sqrt(x * x + y * y + z * z)
    irx, iry, irz = x / rn, y / rn, z / rn
    s5 = sqrt(mpf(5))
    barP20 = s5 * (3 * irz * irz - 1) / 2
    barP20d = s5 * 3 * irz
    if rn < RB:
        rn = RB
    g_ir = -MU / rn ** 2 * (1 + barC20 * (RA / rn) ** 2 * (3 * barP20 + irz * barP20d))
    g_iz = MU / rn ** 2 * (RA / rn) ** 2 * barC20 * barP20d
    return [g_ir * irx, g_ir * iry, g_ir * irz + g_iz]


def air_velocity(r, v, t, wind, alt_shift=0):
    """the velocity relative to the air in ECI axes and the geopotential altitude of the look-ups: src/pybind_dynamics.cpp:43-53 and
    src/wrapper_utils.hpp:93-100,165-172 (the same calls) on exact mpf inputs -> ([3], h)"""
    _, _, alt = geodetic(*r)                                   # the ECI position as if ECEF
    alt = alt + alt_shift
    h = R0 * alt / (R0 + alt) if alt < 86000 else alt          # src/Air.cpp:47-54
    c, s = cos(OMEGA * t), sin(OMEGA * t)
    vg = [v[0] + OMEGA * r[1], v[1] - OMEGA * r[0], v[2]]     # src/Coordinate.cpp:69-73
    vecef = [vg[0] * c + vg[1] * s, -vg[0] * s + vg[1] * c, vg[2]]
    wned = [interp(h, wind[0], wind[1]), interp(h, wind[0], wind[2]), mpf(0)]   # src/wrapper_utils.hpp:82-87
    pe = [r[0] * c + r[1] * s, -r[0] * s + r[1] * c, r[2]]    # eci2ecef, :51-59
    lat, lon, _ = geodetic(*pe)
    cl, sl_, cp, sp = cos(lon / 2), sin(lon / 2), cos(lat / 2), sin(lat / 2)
    q_e2n = [cl * (cp - sp) / sqrt(mpf(2)), sl_ * (cp + sp) / sqrt(mpf(2)), -cl * (cp + sp) / sqrt(mpf(2)), sl_ * (cp - sp) / sqrt(mpf(2))]
    q_i2e = [cos(OMEGA * t / 2), mpf(0), mpf(0), sin(OMEGA * t / 2)]
    q_n2i = conj(quatmult(q_i2e, q_e2n))                       # :104-110
    weci = quatrot(q_n2i, wned)
    va = [vecef[0] * c - vecef[1] * s - weci[0], vecef[0] * s + vecef[1] * c - weci[1], vecef[2] - weci[2]]
    return va, h


def rhs_air(m_e, r_e, v_e, q, t, thrust, area, nozzle, wind, ca, units, barC20, alt_shift=0):
    """src/pybind_dynamics.cpp:30-71 on fp64 inputs given as exact mpf; returns acc / unit_vel (3 mpf).  alt_shift (m) is added
    to the altitude before the atmosphere and wind look-ups: only used to form d(acc)/d(altitude) for the noise bound."""
    um, up, uv = units
    # :33-35 -- the scaling is an fp64 product in the reference (Eigen array * double): the chain starts from those ROUNDED values
    m = f64(float(m_e) * float(um))
    r = [f64(float(c) * float(up)) for c in r_e]
    v = [f64(float(c) * float(uv)) for c in v_e]
    va, h = air_velocity(r, v, t, wind, alt_shift)            # :43-53
    T, P, rho, a = atmosphere(h)
    vn = sqrt(va[0] ** 2 + va[1] ** 2 + va[2] ** 2)
    cav = interp(vn / a, ca[0], ca[1])
    F = [mpf("0.5") * rho * area * cav * vn * -x for x in va]  # :58-59
    Tt = thrust - nozzle * P                                   # :61
    d = quatrot(conj(q), [mpf(1), mpf(0), mpf(0)])             # :62-63
    g = gravity(r, barC20)
    return [((Tt * d[i] + F[i]) / m + g[i]) / uv for i in range(3)]   # :66-70


def velocity_fd_truth(prob, x, phase, barC20, with_alt_sensitivity=True):
    """Exact values of everything the reference's velocity Jacobian of one AERODYNAMIC phase differences
    (lib/con_dynamics.py:353-480): f_c [n, 3] and the quotient -(f_p - f_c)/dx * (tf - to) * unit_t / 2 for the mass (1),
    position (3), velocity (3) and quaternion (4) sweeps -> dict of arrays [n, 3] / [n, 3, k]; `dfdalt` [n, 3] =
    d(acc / unit_vel)/d(altitude) per metre (for the reference's noise bound); `lat`, `alt` per node."""
    nn = [int(v) for v in prob["num_nodes"]]
    S, N = len(nn), sum(nn)
    M = N + S
    um, up, uv, uu, ut = [f64(u) for u in prob["units"]]
    units = (um, up, uv)
    dx = float(prob["dx"])
    ua = sum(nn[:phase])
    xa = ua + phase
    n = nn[phase]
    xm, xr, xv, xq = x[:M], x[M:4 * M].reshape(-1, 3), x[4 * M:7 * M].reshape(-1, 3), x[7 * M:11 * M].reshape(-1, 4)
    xt = x[11 * M + 2 * N:]
    to, tf = float(xt[phase]), float(xt[phase + 1])
    tau = np.asarray(prob["tau"][phase], dtype=np.float64)
    tn = tau * (tf - to) / 2 + (tf + to) / 2                   # fp64, as lib/SectionParameters.py:77-81 forms it
    wt, ct = np.asarray(prob["wind_table"]), np.asarray(prob["ca_table"])
    wind = [[f64(v) for v in wt[:, c]] for c in range(3)]
    ca = [[f64(v) for v in ct[:, c]] for c in range(2)]
    thrust, area, nozzle = f64(prob["thrust"][phase]), f64(prob["reference_area"][phase]), f64(prob["nozzle_area"][phase])
    scale = (f64(tf) - f64(to)) * ut / 2
    out = {"fc": np.zeros((n, 3)), "mass": np.zeros((n, 3)), "position": np.zeros((n, 3, 3)), "velocity": np.zeros((n, 3, 3)),
           "quaternion": np.zeros((n, 3, 4)), "dfdalt": np.zeros((n, 3)), "lat": np.zeros(n), "alt": np.zeros(n)}
    for j in range(n):
        k = xa + 1 + j
        base = dict(m=float(xm[k]), r=[float(v) for v in xr[k]], v=[float(v) for v in xv[k]], q=[float(v) for v in xq[k]])

        def f(**over):
            a = dict(base)
            a.update(over)
            return rhs_air(f64(a["m"]), [f64(v) for v in a["r"]], [f64(v) for v in a["v"]], [f64(v) for v in a["q"]], f64(tn[j]),
                           thrust, area, nozzle, wind, ca, units, barC20, alt_shift=over.get("alt_shift", 0))

        fc = f()

        def quot(fp):
            return [float(-(fp[i] - fc[i]) / f64(dx) * scale) for i in range(3)]

        out["fc"][j] = [float(v) for v in fc]
        out["mass"][j] = quot(f(m=base["m"] + dx))            # the fp64 sum the reference forms (`+= dx`)
        for c in range(3):
            rp = list(base["r"]); rp[c] = rp[c] + dx
            out["position"][j, :, c] = quot(f(r=rp))
            vp = list(base["v"]); vp[c] = vp[c] + dx
            out["velocity"][j, :, c] = quot(f(v=vp))
        for c in range(4):
            qp = list(base["q"]); qp[c] = qp[c] + dx
            out["quaternion"][j, :, c] = quot(f(q=qp))
        if with_alt_sensitivity:
            fs = rhs_air(f64(base["m"]), [f64(v) for v in base["r"]], [f64(v) for v in base["v"]], [f64(v) for v in base["q"]],
                         f64(tn[j]), thrust, area, nozzle, wind, ca, units, barC20, alt_shift=mpf("1e-3"))
            out["dfdalt"][j] = [float((fs[i] - fc[i]) / mpf("1e-3")) for i in range(3)]
        lat, _, alt = geodetic(*[f64(float(v) * float(up)) for v in base["r"]])
        out["lat"][j], out["alt"][j] = float(lat), float(alt)
    return out


# ---- aero path constraints ----
def aero_point(r_e, v_e, q, t_e, wind, up, uv, ut, alt_shift=0):
    """lib/con_aero.py:39-87 (scale in fp64) + src/wrapper_utils.hpp:89-111,163-175 -> (angle of attack [rad], dynamic pressure [Pa])
    as exact mpf of fp64 inputs"""
    r = [f64(float(c) * float(up)) for c in r_e]
    v = [f64(float(c) * float(uv)) for c in v_e]
    t = f64(float(t_e) * float(ut))
    va, h = air_velocity(r, v, t, wind, alt_shift)
    nv = sqrt(va[0] ** 2 + va[1] ** 2 + va[2] ** 2)
    d = quatrot(conj(q), [mpf(1), mpf(0), mpf(0)])             # :91-92
    nd = sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    rho = atmosphere(h)[2]
    qdyn = mpf("0.5") * rho * nv * nv
    if nv < mpf("1e-6"):
        return mpf(0), qdyn
    c = sum((va[i] / nv) * (d[i] / nd) for i in range(3))
    from mpmath import acos
    return (mpf(0) if c > 1 else acos(c)), qdyn


def aero_fd_truth(prob, x, spec):
    """Exact values of what lib/con_aero.py:311-471 differences, for the nodes of `spec` (rows of (phase, range_all)): per node
    alpha, q and the quotients (f_p - f_c)/dx of BOTH (f = alpha [rad], q [Pa]: the caller forms a kind's gradient from them, the
    product rule being exact for the quotient of q * alpha: see tests/fd_noise.py) for the position (3), velocity (3) and
    quaternion (4) sweeps from the UNMODIFIED node (no `+= dx, -= dx` drift); the t0 / tf quotients are exactly zero (the
    air-relative velocity does not depend on the Earth angle) and are returned as computed, to show it.
    -> dict: alpha [R], q [R], d_alpha [R, 12], d_q [R, 12] (columns: position xyz, velocity xyz, quaternion wxyz, t0, tf),
    dalpha_dalt, dq_dalt [R] per metre, lat, alt [R]"""
    nn = [int(v) for v in prob["num_nodes"]]
    S, N = len(nn), sum(nn)
    M = N + S
    up, uv, ut = [f64(prob["units"][k]) for k in (1, 2, 4)]
    dx = float(prob["dx"])
    xr, xv, xq = x[M:4 * M].reshape(-1, 3), x[4 * M:7 * M].reshape(-1, 3), x[7 * M:11 * M].reshape(-1, 4)
    xt = x[11 * M + 2 * N:]
    wt = np.asarray(prob["wind_table"])
    wind = [[f64(v) for v in wt[:, c]] for c in range(3)]
    rows = []
    for ph, all_nodes in spec:
        ph = int(ph)
        xa = sum(nn[:ph]) + ph
        to, tf = float(xt[ph]), float(xt[ph + 1])
        tau = np.asarray(prob["tau"][ph], dtype=np.float64)

        def tnode(k, a, b):                                    # lib/SectionParameters.py:77-81 in fp64
            return a if k == 0 else float(tau[k - 1] * (b - a) / 2 + (b + a) / 2)

        for k in range(nn[ph] + 1 if all_nodes else 1):
            r0 = [float(v) for v in xr[xa + k]]; v0 = [float(v) for v in xv[xa + k]]; q0 = [float(v) for v in xq[xa + k]]

            def f(r=r0, v=v0, q=q0, t=tnode(k, to, tf), alt_shift=0):
                return aero_point(r, v, [f64(c) for c in q], t, wind, up, uv, ut, alt_shift)

            ac, qc = f()
            cols = []
            for c in range(3):
                rp = list(r0); rp[c] = rp[c] + dx
                cols.append(f(r=rp))
            for c in range(3):
                vp = list(v0); vp[c] = vp[c] + dx
                cols.append(f(v=vp))
            for c in range(4):
                qp = list(q0); qp[c] = qp[c] + dx
                cols.append(f(q=qp))
            cols.append(f(t=tnode(k, to + dx, tf)))
            cols.append(f(t=tnode(k, to, tf + dx)))
            ash, qsh = f(alt_shift=mpf("1e-3"))
            lat, _, alt = geodetic(*[f64(float(c) * float(up)) for c in r0])
            rows.append((float(ac), float(qc), [float((a - ac) / f64(dx)) for a, _ in cols], [float((b - qc) / f64(dx)) for _, b in cols],
                         float((ash - ac) / mpf("1e-3")), float((qsh - qc) / mpf("1e-3")), float(lat), float(alt)))
    keys = ("alpha", "q", "d_alpha", "d_q", "dalpha_dalt", "dq_dalt", "lat", "alt")
    return {k: np.array([row[i] for row in rows]) for i, k in enumerate(keys)}
