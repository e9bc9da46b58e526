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

The "downrange" rows: values as con_waypoint.py:531-534,551-554,742,771-778 (the `max` row divides by the `min` bound,
:778); Jacobian = downrange_gradient (:583-607) scaled like the value, position entries on the position list and the t
entry on the t list -- the form the reference intended.  The reference's own lists are scrambled (it appends the t VALUE
to the position values, :702-706,915-919,932-936, and scales the `max` row's t entry by the `max` bound):
`reference_downrange_lists()` restates that scramble, so that the fixture written from the reference pins every number.
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


# ---------------------------------------------------------------- lib/downrange.py:32-111
def distance_vincenty(lat_o, lon_o, lat_t, lon_t):    # lib/downrange.py:32-111
    Ra = 6378137.0
    f = 1.0 / 298.257223563
    Rb = Ra * (1.0 - f)
    lat1, lon1 = lat_o * math.pi / 180.0, lon_o * math.pi / 180.0
    lat2, lon2 = lat_t * math.pi / 180.0, lon_t * math.pi / 180.0
    if lon2 - lon1 == 0.0:
        return 0.0
    U1 = math.atan((1.0 - f) * math.tan(lat1))
    U2 = math.atan((1.0 - f) * math.tan(lat2))
    dl = lon2 - lon1
    lam = dl
    sin_sigma = cos_sigma = sigma = cos_alpha = cos_2sm = 0.0
    for _ in range(5000):
        sin_sigma = math.sqrt((math.cos(U2) * math.sin(lam)) ** 2 +
                              (math.cos(U1) * math.sin(U2) - math.sin(U1) * math.cos(U2) * math.cos(lam)) ** 2)
        cos_sigma = math.sin(U1) * math.sin(U2) + math.cos(U1) * math.cos(U2) * math.cos(lam)
        sigma = math.atan2(sin_sigma, cos_sigma)
        sin_alpha = math.cos(U1) * math.cos(U2) * math.sin(lam) / sin_sigma
        cos_alpha = math.sqrt(1.0 - sin_alpha ** 2)
        cos_2sm = cos_sigma - 2.0 * math.sin(U1) * math.sin(U2) / cos_alpha ** 2
        coeff = f / 16.0 * cos_alpha ** 2 * (4.0 + f * (4.0 - 3.0 * cos_alpha ** 2))
        prev = lam
        lam = dl + (1.0 - coeff) * f * sin_alpha * (
            sigma + coeff * sin_sigma * (cos_2sm + coeff * cos_sigma * (-1.0 + 2.0 * cos_2sm)))
        if abs(lam - prev) < 1e-12:
            break
    u2 = cos_alpha ** 2 * (Ra ** 2 - Rb ** 2) / Rb ** 2
    A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)))
    B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)))
    ds = B * sin_sigma * (cos_2sm + 0.25 * B * (cos_sigma * (-1.0 + 2.0 * cos_2sm ** 2) -
                                                (1.0 / 6.0) * B * cos_2sm * (-3.0 + 4.0 * sin_sigma ** 2) * (-3.0 + 4.0 * cos_2sm ** 2)))
    return Rb * A * (sigma - ds)


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


def _f_downrange(p_, v_, t_, sp, origin):             # con_waypoint.py:590-598
    llh = eci2geodetic(p_ * sp["units"]["position"], t_ * sp["units"]["t"])
    return np.array([distance_vincenty(origin[0], origin[1], llh[0], llh[1])])


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
            for comp, key, scale in ((0, "lat", 90.0), (1, "lon", 180.0), (2, "altitude", None), (0, "downrange", None)):
                for kind in ("exact", "min", "max"):
                    if key in w and kind in w[key]:
                        grp = "eqpos" if kind == "exact" else "ineqpos"
                        if key == "downrange":
                            lc = pdict["LaunchCondition"]             # :533-534; max reads the MIN bound (:778)
                            bound = float(w[key]["min" if kind == "max" else kind])
                            rows.append((grp, i, sp["xa"][i], "dr", 0, kind, bound, None, (float(lc["lat"]), float(lc["lon"]))))
                        else:
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


_FN = {"llh": _f_llh, "iip": _f_iip, "elev": _f_elev, "dr": _f_downrange}
_VARS = {"llh": ("position", "t"), "iip": ("position", "velocity", "t"), "elev": ("position", "t"), "dr": ("position", "t")}


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


def reference_downrange_lists(J, rows, group, condition_bounds):
    """The value lists the reference itself returns for a position group that contains downrange rows (module docstring):
    jac["position"]["coo"][2] takes, after the three position values of a downrange row, that row's t value as a fourth
    (:702-706,915-919,932-936), jac["t"]["coo"][2] gets none; the `max` row's t value is scaled by -1/max where its
    position values (and ours) are scaled by -1/min.  condition_bounds[row index] = (min, max) of a max row.
    -> (position values, t values) as flat arrays in the reference's order."""
    mine = [r for r in rows if r[0] == group]
    pv, tv = [], []
    for ir, r in enumerate(mine):
        pos3 = list(J["position"][2][3 * ir:3 * ir + 3])
        tval = J["t"][2][ir]
        if r[3] == "dr":
            if r[5] == "max":
                lo, hi = condition_bounds[ir]
                tval = tval * lo / hi
            pv.extend(pos3 + [tval])
        else:
            pv.extend(pos3)
            tv.append(tval)
    return np.array(pv), np.array(tv)
