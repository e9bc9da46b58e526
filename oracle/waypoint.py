"""CPU ORACLE (test infrastructure, NOT the product) for the waypoint rows of SURVEY.md 8f row f-4:
lib/con_waypoint.py -- geodetic position, instantaneous impact point and antenna elevation at the first state node of
named sections, with their forward-difference Jacobians.

A numpy / math restatement of the reference algorithm, one decision vector per call; every function cites the reference
lines it follows.  Pinned against the reference itself: tests/golden/g13_waypoint.npz was written by
tests/golden/make_golden.py from the imported reference modules (lib.con_waypoint over lib.coordinate, lib.IIP);
tests/test_waypoint.py checks this file against it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it.

The reference's gradient helpers perturb VIEWS of xdict in place (`pos_[j] += dx ... pos_[j] -= dx`,
con_waypoint.py:58-61,224-232,570-573), and (p + dx) - dx is not always p: after the first row of a node the state
the later rows (and later callbacks) see has drifted by an ulp.  `drift=True` restates exactly that, to pin the oracle
bit for bit; `drift=False` (the product's semantics: x is never mutated, every row differences the same centre) agrees
with it to forward-difference noise.

The "downrange" rows are not restated: every one of their Jacobian blocks appends the t entry to the position list
(con_waypoint.py:702-706,915-919,932-936), so the reference cannot assemble them.
"""
import math

import numpy as np

from .knot_terminal import split

OMEGA = 7.2921151467e-5     # lib/coordinate.py:228
A_E = 6378137.0
F_E = 1.0 / 298.257223563
MU = 3.986004418e14


# ---------------------------------------------------------------- lib/coordinate.py
def eci2ecef(p, t):                                   # :217-237
    c, s = math.cos(OMEGA * t), math.sin(OMEGA * t)
    return np.array([p[0] * c + p[1] * s, -p[0] * s + p[1] * c, p[2]])


def vel_eci2ecef(v, p, t):                            # :261-278
    rot = np.cross(np.array([0, 0, OMEGA]), p)
    return eci2ecef(v - rot, t)


def ecef2geodetic(x, y, z):                           # :103-128
    a = A_E
    b = a * (1.0 - F_E)
    e2 = (a ** 2 - b ** 2) / a ** 2
    ep2 = (a ** 2 - b ** 2) / b ** 2
    p = math.sqrt(x ** 2 + y ** 2)
    theta = math.atan2(z * a, p * b)
    lat = math.atan2(z + ep2 * b * math.sin(theta) ** 3, p - e2 * a * math.cos(theta) ** 3)
    lon = math.atan2(y, x)
    N = a / math.sqrt(1.0 - e2 * math.sin(lat) ** 2)
    alt = p / math.cos(lat) - N
    return np.array((math.degrees(lat), math.degrees(lon), alt))


def geodetic2ecef(lat, lon, alt):                     # :131-153
    a = A_E
    b = a * (1.0 - F_E)
    e2 = (a ** 2 - b ** 2) / a ** 2
    N = a / math.sqrt(1.0 - e2 * math.sin(math.radians(lat)) ** 2)
    x = (N + alt) * math.cos(math.radians(lat)) * math.cos(math.radians(lon))
    y = (N + alt) * math.cos(math.radians(lat)) * math.sin(math.radians(lon))
    z = (N * (1 - e2) + alt) * math.sin(math.radians(lat))
    return np.array((x, y, z))


def eci2geodetic(p, t):                               # :573-588
    e = eci2ecef(p, t)
    return ecef2geodetic(e[0], e[1], e[2])


def _quatmult(q, p):                                  # :31-37
    return np.array([q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3],
                     q[1] * p[0] + q[0] * p[1] - q[3] * p[2] + q[2] * p[3],
                     q[2] * p[0] + q[3] * p[1] + q[0] * p[2] - q[1] * p[3],
                     q[3] * p[0] - q[2] * p[1] + q[1] * p[2] + q[0] * p[3]])


def _conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def antenna_vertical(pos_ecef):
    """quatrot(quat_nedg2ecef(pos), [0, 0, -1]) (:55-68,335-371): the local vertical of a ground point, in ECEF."""
    la, lo, _ = ecef2geodetic(pos_ecef[0], pos_ecef[1], pos_ecef[2])
    p, l = math.radians(la), math.radians(lo)
    c_hl, s_hl, c_hp, s_hp = math.cos(l / 2.0), math.sin(l / 2.0), math.cos(p / 2.0), math.sin(p / 2.0)
    q = np.array([c_hl * (c_hp - s_hp) / math.sqrt(2.0), s_hl * (c_hp + s_hp) / math.sqrt(2.0),
                  -c_hl * (c_hp + s_hp) / math.sqrt(2.0), s_hl * (c_hp - s_hp) / math.sqrt(2.0)])
    qc = _conj(q)                                     # nedg -> ecef
    vq = np.array((0.0, 0.0, 0.0, -1.0))
    return _quatmult(_conj(qc), _quatmult(vq, qc))[1:4]


# ---------------------------------------------------------------- lib/IIP.py:30-135
def posLLH_IIP_FAA(pe, ve, n_iter=5):
    a = 6378137
    b = a * (1.0 - F_E)
    e2 = 2.0 * F_E - F_E * F_E
    none = np.zeros(3)
    r_k1 = b
    r0 = np.linalg.norm(pe)
    if r0 < r_k1:
        return none
    vi = ve + np.cross(np.array([0.0, 0.0, OMEGA]), pe)
    v0 = np.linalg.norm(vi)
    eps_cos = (r0 * v0 ** 2 / MU) - 1
    if eps_cos >= 1:
        return none
    a_t = r0 / (1 - eps_cos)
    eps_sin = np.dot(pe, vi) / math.sqrt(MU * a_t)
    eps2 = eps_cos ** 2 + eps_sin ** 2
    if (math.sqrt(eps2) <= 1) and (a_t * (1 - math.sqrt(eps2)) - a >= 0):
        return none
    for _ in range(n_iter):
        eps_k_cos = (a_t - r_k1) / a_t
        if eps2 - eps_k_cos ** 2 < 0:
            return none
        eps_k_sin = -math.sqrt(eps2 - eps_k_cos ** 2)
        d_cos = (eps_k_cos * eps_cos + eps_k_sin * eps_sin) / eps2
        d_sin = (eps_k_sin * eps_cos - eps_k_cos * eps_sin) / eps2
        fs = (d_cos - eps_cos) / (1 - eps_cos)
        gs = (d_sin + eps_sin - eps_k_sin) * math.sqrt(a_t ** 3 / MU)
        Ek = fs * pe[0] + gs * vi[0]
        Fk = fs * pe[1] + gs * vi[1]
        Gk = fs * pe[2] + gs * vi[2]
        r_k2 = a / math.sqrt((e2 / (1 - e2)) * (Gk / r_k1) ** 2 + 1)
        r_prev = r_k1
        r_k1 = r_k2
    if abs(r_prev - r_k2) > 1:
        return none
    delta = np.arctan2(d_sin, d_cos)
    time_sec = (delta + eps_sin - eps_k_sin) * math.sqrt(a_t ** 3 / MU)
    phi = np.arctan2(np.tan(np.arcsin(Gk / r_k2)), 1 - e2)
    lam = np.arctan2(Fk, Ek) - OMEGA * time_sec
    return np.array([phi, lam, 0.0]) * 180.0 / np.pi


# ---------------------------------------------------------------- the three node functions of the rows
def _f_llh(p_, v_, t_, sp, _ant):
    return eci2geodetic(p_ * sp["units"]["position"], t_ * sp["units"]["t"])


def _f_iip(p_, v_, t_, sp, _ant):
    pos, to = p_ * sp["units"]["position"], t_ * sp["units"]["t"]
    return posLLH_IIP_FAA(eci2ecef(pos, to), vel_eci2ecef(v_ * sp["units"]["velocity"], pos, to))


def _f_elev(p_, v_, t_, sp, ant):                     # con_waypoint.py:45-51
    pe = eci2ecef(p_ * sp["units"]["position"], t_ * sp["units"]["t"])
    d = pe - ant
    d = d / np.linalg.norm(d)
    return np.array([np.dot(d, antenna_vertical(ant))])


def make_rows(sp, pdict, condition):
    """Row descriptors of the five groups, in the reference's emission order.  Each row:
    (group, section, node, function, component, kind, bound, scale, antenna_ecef | None) with kind in
    exact / min / max; value and Jacobian scaling as con_waypoint.py writes them."""
    rows = []
    S = sp["S"]
    names = [pdict["params"][i]["name"] for i in range(S)]
    wp = condition.get("waypoint")
    if wp is not None:
        for i in range(S - 1):
            if names[i] not in wp:
                continue
            w = wp[names[i]]
            if "downrange" in w:
                raise NotImplementedError("downrange rows: the reference cannot assemble their Jacobian")
            for comp, key, scale in ((0, "lat", 90.0), (1, "lon", 180.0), (2, "altitude", None)):
                for kind in ("exact", "min", "max"):
                    if key in w and kind in w[key]:
                        grp = "eqpos" if kind == "exact" else "ineqpos"
                        rows.append((grp, i, sp["xa"][i], "llh", comp, kind, float(w[key][kind]), scale, None))
            for comp, key, scale in ((0, "lat_IIP", 90.0), (1, "lon_IIP", 180.0)):
                for kind in ("exact", "min", "max"):
                    if key in w and kind in w[key]:
                        grp = "eqiip" if kind == "exact" else "ineqiip"
                        rows.append((grp, i, sp["xa"][i], "iip", comp, kind, float(w[key][kind]), scale, None))
    for ant in (condition.get("antenna") or {}).values():
        ecef = geodetic2ecef(ant["lat"], ant["lon"], ant["altitude"])
        for i in range(S - 1):
            if names[i] in ant["elevation_min"]:
                rows.append(("antenna", i, sp["xa"][i], "elev", 0, "min",
                             math.sin(ant["elevation_min"][names[i]] * np.pi / 180.0), None, ecef))
    return rows


_FN = {"llh": _f_llh, "iip": _f_iip, "elev": _f_elev}
_VARS = {"llh": ("position", "t"), "iip": ("position", "velocity", "t"), "elev": ("position", "t")}


def _value(f, row):
    _, _, _, fn, comp, kind, bound, scale, _ = row
    if fn == "elev":
        return f[0] - bound                                           # :97
    if scale is None:                                                 # altitude: ratio to the bound (:549,766,769)
        return -(f[comp] / bound) + 1.0 if kind == "max" else (f[comp] / bound) - 1.0
    if kind == "max":                                                 # :364,374,751,759
        return -(f[comp] - bound) / scale
    return (f[comp] - bound) / scale


def values(x, sp, rows, group):
    """equality_posLLH / inequality_posLLH / equality_IIP / inequality_IIP / inequality_antenna -> 1-D array or None"""
    _, pos, vel, _, _, t = split(np.asarray(x, dtype=np.float64), sp["M"], sp["N"])
    out = [_value(_FN[r[3]](pos[r[2]], vel[r[2]], t[r[1]], sp, r[8]), r) for r in rows if r[0] == group]
    return np.array(out) if out else None


def jacobian(x, sp, rows, group, drift=False):
    """The matching *_jac_* function -> {var: (rows, cols, vals, shape)} or None.  drift=True mutates a private copy of x
    the way the reference mutates xdict (module docstring) and returns that copy as the second result."""
    mine = [r for r in rows if r[0] == group]
    x = np.array(x, dtype=np.float64)
    if not mine:
        return (None, x) if drift else None
    M, S, dx = sp["M"], sp["S"], sp["dx"]
    _, pos, vel, _, _, t = split(x, M, sp["N"])                       # views of the private copy
    fn = mine[0][3]
    out = {k: ([], [], []) for k in _VARS[fn]}
    for ir, r in enumerate(mine):
        _, sec, node, fn, comp, kind, bound, scale, ant = r
        f = _FN[fn]
        p_ = pos[node] if drift else pos[node].copy()
        v_ = vel[node] if drift else vel[node].copy()
        t_ = t[sec]
        fc = f(p_, v_, t_, sp, ant)
        gp, gv = np.zeros((len(fc), 3)), np.zeros((len(fc), 3))
        for j in range(3):                                            # :224-237 (position j, then velocity j), :570-574, :57-61
            keep = p_[j]
            p_[j] += dx
            gp[:, j] = (f(p_, v_, t_, sp, ant) - fc) / dx
            p_[j] -= dx
            if not drift:
                p_[j] = keep
            if fn == "iip":
                keep = v_[j]
                v_[j] += dx
                gv[:, j] = (f(p_, v_, t_, sp, ant) - fc) / dx
                v_[j] -= dx
                if not drift:
                    v_[j] = keep
        gt = (f(p_, v_, t_ + dx, sp, ant) - fc) / dx
        if fn == "elev":
            sc = lambda g: g                                          # noqa: E731  (:147-153)
        else:
            den = bound if scale is None else scale
            sc = (lambda g: -g / den) if kind == "max" else (lambda g: g / den)
        out["position"][0].extend([ir] * 3); out["position"][1].extend(range(3 * node, 3 * node + 3))
        out["position"][2].extend(sc(gp[comp, :]))
        if fn == "iip":
            out["velocity"][0].extend([ir] * 3); out["velocity"][1].extend(range(3 * node, 3 * node + 3))
            out["velocity"][2].extend(sc(gv[comp, :]))
        out["t"][0].append(ir); out["t"][1].append(sec); out["t"][2].append(sc(gt[comp]))
    width = {"position": 3 * M, "velocity": 3 * M, "t": S + 1}
    res = {k: (np.array(v[0], dtype=np.int32), np.array(v[1], dtype=np.int32), np.array(v[2], dtype=np.float64),
               (len(mine), width[k])) for k, v in out.items()}
    return (res, x) if drift else res
