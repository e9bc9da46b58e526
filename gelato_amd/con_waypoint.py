"""Waypoint, impact-point and antenna constraints on the GPU.

Drop-in for the reference's lib/con_waypoint.py: the same ten functions with the same
``fn(xdict, pdict, unitdict, condition)`` signature and the same return layouts

  equality_posLLH / inequality_posLLH / equality_IIP / inequality_IIP / inequality_antenna   -> 1-D ndarray or None
  equality_jac_* / inequality_jac_*   -> {var: {"coo": [rows i4, cols i4, vals f8], "shape": (r, c)}} or None

(lib/con_waypoint.py:70-105,108-161,164-207,243-327,330-381,384-504,507-560,611-714,717-784,787-945).

Every row is a node-function row of the device table the init / knot / terminal rows live in (gel_rows_configure): a
function of ONE knot state and its knot time -- geodetic latitude / longitude / altitude (fn 9-11), latitude /
longitude of the instantaneous impact point (fn 12-13, FAA algorithm of lib/IIP.py), sine of the elevation seen from a
ground antenna (fn 14) -- with its forward difference over position, velocity and knot time formed in the kernel, and
the reference's scaling of value and difference.  One launch per callback evaluates all of them together with the other
rows; the functions here slice the shared frame.  xdict is never mutated (the reference perturbs views of it in place).

The "downrange" rows (Vincenty distance from the launch point, fn 15): values exactly as the reference computes them
(con_waypoint.py:531-534,551-554,742,771-778 -- including its `max` row, which divides by the `min` bound, :778, and
therefore needs one).  Their Jacobian blocks come in the form the reference intended: downrange_gradient (:583-607)
scaled like the value (`/ exact`, `/ min`, `-... / min`), position entries on the position list and the t entry on the
t list.  The reference itself appends the t VALUE to the position list (:702-706,915-919,932-936: four values for three
index pairs, none for the t pair) and scales the `max` row's t entry by the `max` bound although its value uses `min`;
tests/test_waypoint.py unscrambles its lists to check every number against the reference's own output.
"""
import math

import numpy as np

from . import con_init_terminal_knot as _ck
from .engine import Engine

_A = 6378137.0
_F = 1.0 / 298.257223563
_GROUPS = ("eqpos", "ineqpos", "eqiip", "ineqiip", "antenna")


def _geodetic2ecef(lat, lon, alt):
    """lib/coordinate.py:131-153"""
    b = _A * (1.0 - _F)
    e2 = (_A ** 2 - b ** 2) / _A ** 2
    sl, cl = math.sin(math.radians(lat)), math.cos(math.radians(lat))
    n = _A / math.sqrt(1.0 - e2 * sl ** 2)
    return ((n + alt) * cl * math.cos(math.radians(lon)), (n + alt) * cl * math.sin(math.radians(lon)),
            (n * (1 - e2) + alt) * sl)


def _vertical(ecef):
    """The local vertical at a ground point, quatrot(quat_nedg2ecef(p), [0, 0, -1]) of con_waypoint.py:50: the outward
    ellipsoid normal (cos lat cos lon, cos lat sin lon, sin lat) at the point's geodetic latitude / longitude
    (lib/coordinate.py:103-128)."""
    x, y, z = ecef
    b = _A * (1.0 - _F)
    e2 = (_A ** 2 - b ** 2) / _A ** 2
    ep2 = (_A ** 2 - b ** 2) / b ** 2
    p = math.sqrt(x ** 2 + y ** 2)
    th = math.atan2(z * _A, p * b)
    lat = math.atan2(z + ep2 * b * math.sin(th) ** 3, p - e2 * _A * math.cos(th) ** 3)
    lon = math.atan2(y, x)
    return (math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat))


def build_rows(pdict, condition, xa):
    """The node-function rows of the five groups in the reference's emission order:
    [(group, section, node, long-form row of Engine.rows_configure)]."""
    S = pdict["num_sections"]
    names = [pdict["params"][i]["name"] for i in range(S)]
    SH, RAW, NEG = Engine.MODE_SHIFTED, Engine.MODE_RAW_DIFFERENCE, Engine.MODE_NEGATED
    out = []
    wp = condition.get("waypoint")
    if wp is not None:
        for i in range(S - 1):                                     # the last section's start is not looked at (:180,523)
            if names[i] not in wp:
                continue
            w = wp[names[i]]
            specs = (("latitude_deg", "lat", 90.0, "pos"), ("longitude_deg", "lon", 180.0, "pos"),
                     ("altitude", "altitude", None, "pos"), ("downrange", "downrange", None, "pos"),
                     ("lat_IIP_deg", "lat_IIP", 90.0, "iip"), ("lon_IIP_deg", "lon_IIP", 180.0, "iip"))
            for f, key, scale, fam in specs:
                if key not in w:
                    continue
                for kind in ("exact", "min", "max"):
                    if kind not in w[key]:
                        continue
                    bound = float(w[key][kind])
                    grp = ("eq" if kind == "exact" else "ineq") + fam
                    neg = NEG if kind == "max" else 0
                    if f == "downrange":                            # downrange / bound - 1 (:554,774); max: -(downrange / MIN) + 1 (:778)
                        if kind == "max":
                            bound = float(w[key]["min"])            # KeyError without a min bound, as in the reference
                        lc = pdict["LaunchCondition"]               # :533-534
                        row = (f, xa[i], i, RAW | neg, [bound, 1.0, float(lc["lat"]), float(lc["lon"])])
                    elif scale is None:                             # f / bound - 1 (:549,766,769); difference / bound
                        row = (f, xa[i], i, RAW | neg, [bound, 1.0])
                    else:                                           # (f - bound) / scale (:539,544,749-759)
                        row = (f, xa[i], i, SH | RAW | neg, [scale, bound])
                    out.append((grp, i, xa[i], row))
    for ant in (condition.get("antenna") or {}).values():
        ecef = _geodetic2ecef(ant["lat"], ant["lon"], ant["altitude"])
        up = _vertical(ecef)
        for i in range(S - 1):
            if names[i] in ant["elevation_min"]:                    # sin(elevation) - sin(elevation_min) (:92-97)
                smin = math.sin(ant["elevation_min"][names[i]] * math.pi / 180.0)
                out.append(("antenna", i, xa[i], ("sin_elevation", xa[i], i, RAW, [1.0, smin, *ecef, *up])))
    # the callers slice by group: keep each group contiguous, reference order inside
    return [r for g in _GROUPS for r in out if r[0] == g]


def _group(xdict, pdict, unitdict, condition, group, need_jac):
    R = _ck.rows_of(pdict, unitdict, condition)
    a, b = R.waypoint_slices[group]
    if a == b:
        return None, None, None
    con, jfn = R.evaluate(xdict, pdict, need_jac)
    k0 = R.nlin + R.waypoint_base
    meta = R.waypoint_rows[a:b]
    return con[k0 + a:k0 + b].copy(), (jfn[R.waypoint_base + a:R.waypoint_base + b] if need_jac else None), meta


def _jac(xdict, pdict, unitdict, condition, group, with_velocity):
    _, jfn, meta = _group(xdict, pdict, unitdict, condition, group, True)
    if jfn is None:
        return None
    n, M, S = len(meta), pdict["M"], pdict["num_sections"]
    # the index arrays depend on the row table only: formed once per table and group (the values are fresh arrays per call)
    R = _ck.rows_of(pdict, unitdict, condition)
    pat = R.__dict__.setdefault("waypoint_pattern", {}).get(group)
    if pat is None:
        pat = R.waypoint_pattern[group] = (
            np.repeat(np.arange(n, dtype=np.int32), 3), np.array([3 * node + c for (_, _, node, _) in meta for c in range(3)], dtype=np.int32),
            np.arange(n, dtype=np.int32), np.array([sec for (_, sec, _, _) in meta], dtype=np.int32))
    rows3, cols3, rows1, secs = pat
    if pdict.get("gelato_amd_share_values"):
        # shared value arrays (rewritten by the next evaluation, like the defect groups'): the blocks are built once per table and
        # group, the values copied into their arrays
        sh = R.__dict__.setdefault("waypoint_shared", {}).get((group, with_velocity))
        if sh is None:
            sh = {"position": {"coo": [rows3, cols3, np.empty(3 * n)], "shape": (n, 3 * M)}}
            if with_velocity:
                sh["velocity"] = {"coo": [rows3, cols3, np.empty(3 * n)], "shape": (n, 3 * M)}
            sh["t"] = {"coo": [rows1, secs, np.empty(n)], "shape": (n, S + 1)}
            R.waypoint_shared[(group, with_velocity)] = sh
        sh["position"]["coo"][2].reshape(n, 3)[:] = jfn[:, 0:3]
        if with_velocity:
            sh["velocity"]["coo"][2].reshape(n, 3)[:] = jfn[:, 3:6]
        sh["t"]["coo"][2][:] = jfn[:, 6]
        return dict(sh)
    # fresh arrays per call: ravel() of a column slice copies only when the group has more than one row (a one-row slice is
    # contiguous and would be a view of the buffer the next callback overwrites)
    jac = {"position": {"coo": [rows3, cols3, np.array(jfn[:, 0:3]).ravel()], "shape": (n, 3 * M)}}
    if with_velocity:
        jac["velocity"] = {"coo": [rows3, cols3, np.array(jfn[:, 3:6]).ravel()], "shape": (n, 3 * M)}
    jac["t"] = {"coo": [rows1, secs, jfn[:, 6].copy()], "shape": (n, S + 1)}
    return jac


def inequality_antenna(xdict, pdict, unitdict, condition):
    """Inequality constraint about antenna elevation angle."""
    return _group(xdict, pdict, unitdict, condition, "antenna", False)[0]


def inequality_jac_antenna(xdict, pdict, unitdict, condition):
    """Jacobian of inequality_antenna."""
    return _jac(xdict, pdict, unitdict, condition, "antenna", False)


def equality_IIP(xdict, pdict, unitdict, condition):
    """Equality constraint about IIP position."""
    return _group(xdict, pdict, unitdict, condition, "eqiip", False)[0]


def equality_jac_IIP(xdict, pdict, unitdict, condition):
    """Jacobian of equality_IIP."""
    return _jac(xdict, pdict, unitdict, condition, "eqiip", True)


def inequality_IIP(xdict, pdict, unitdict, condition):
    """Inequality constraint about IIP position."""
    return _group(xdict, pdict, unitdict, condition, "ineqiip", False)[0]


def inequality_jac_IIP(xdict, pdict, unitdict, condition):
    """Jacobian of inequality_IIP."""
    return _jac(xdict, pdict, unitdict, condition, "ineqiip", True)


def equality_posLLH(xdict, pdict, unitdict, condition):
    """Equality constraint about the geodetic position at knots."""
    return _group(xdict, pdict, unitdict, condition, "eqpos", False)[0]


def equality_jac_posLLH(xdict, pdict, unitdict, condition):
    """Jacobian of equality_posLLH."""
    return _jac(xdict, pdict, unitdict, condition, "eqpos", False)


def inequality_posLLH(xdict, pdict, unitdict, condition):
    """Inequality constraint about the geodetic position at knots."""
    return _group(xdict, pdict, unitdict, condition, "ineqpos", False)[0]


def inequality_jac_posLLH(xdict, pdict, unitdict, condition):
    """Jacobian of inequality_posLLH."""
    return _jac(xdict, pdict, unitdict, condition, "ineqpos", False)
