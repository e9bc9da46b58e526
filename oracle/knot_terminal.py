"""CPU ORACLE (test infrastructure, NOT the product) for SURVEY.md 8f rows f-4 / f-2 / f-3:
the init / time / knot / terminal rows of lib/con_init_terminal_knot.py, the shipped user constraint
(example/user_constraints.py through lib/jac_fd.py) and the from-file initial guess (initialize.py:322-409).

A numpy restatement of the reference algorithm, one decision vector per call; every function cites the
reference lines it follows.  Pinned against the reference itself: tests/golden/g11_knot_terminal.npz and
g12_initial_guess.npz were written by tests/golden/make_golden.py from the imported reference modules
(lib.con_init_terminal_knot, example/user_constraints.py + lib.jac_fd, initialize.py); tests/test_knot_oracle_golden.py
checks this file against them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Inputs are the packed decision vector x = [mass M | position 3M | velocity 3M | quaternion 4M | u 2N | t S+1] and a
plain `spec` dict (see make_spec) holding what the reference reads from pdict / unitdict / condition.
"""
import math

import numpy as np

MU = 3.986004418e14   # src/Earth.cpp:41, lib/coordinate.py:601
RA = 6378137.0


def make_spec(pdict, unitdict, condition):
    """What the five constraint groups read: node counts, knot times and their reference events, jettisoned masses,
    stage ignition / separation sections, units, dx, initial state and terminal targets."""
    S = pdict["num_sections"]
    ps = pdict["ps_params"]
    P = pdict["params"]
    ev = pdict["event_index"]
    stages = []
    for stage in pdict["RocketStage"].values():
        if stage["separation_at"] is not None:
            drop = sum(item["mass"] for item in (stage["dropMass"] or {}).values())
            stages.append((ev[stage["ignition_at"]], ev[stage["separation_at"]],
                           stage["mass_dry"] + stage["mass_propellant"] + drop))
    return {
        "S": S, "nodes": [ps.nodes(i) for i in range(S)], "xa": [ps.index_start_x(i) for i in range(S)],
        "N": pdict["N"], "M": pdict["M"],
        "time": [p["time"] for p in P],
        "time_ref": [ev[p["time_ref"]] if isinstance(p["time_ref"], str) and p["time_ref"] in ev else -1 for p in P],
        "mass_jettison": [p["mass_jettison"] for p in P],
        "stages": stages,
        # lib/con_trajectory.py: ignition / cut-off sections and burnt mass of every stage; attitude option per section
        "burns": [(ev[st["ignition_at"]], ev[st["cutoff_at"]],
                   st["mass_propellant"] + sum(it["mass"] for it in (st["dropMass"] or {}).values()))
                  for st in pdict["RocketStage"].values()],
        "attitude": [p["attitude"] for p in P],
        "ua": [ps.index_start_u(i) for i in range(S)],
        "units": {k: float(unitdict[k]) for k in ("mass", "position", "velocity", "u", "t")},
        "dx": float(pdict["dx"]),
        "payload_mode": condition["OptimizationMode"] == "Payload",
        "init": {k: np.asarray(condition["init"][k], dtype=np.float64) for k in ("mass", "position", "velocity", "quaternion")},
        "terminal": {k: condition.get(k) for k in ("altitude_perigee", "altitude_apogee", "inclination", "radius",
                                                   "vel_tangential_geocentric", "flightpath_vel_inertial_geocentric")},
    }


def split(x, M, N):
    o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N])
    return (x[o[0]:o[1]], x[o[1]:o[2]].reshape(-1, 3), x[o[2]:o[3]].reshape(-1, 3), x[o[3]:o[4]].reshape(-1, 4),
            x[o[4]:o[5]].reshape(-1, 2), x[o[5]:])


def _coo(rows, cols, vals, shape):
    return {"coo": [np.asarray(rows, dtype=np.int32), np.asarray(cols, dtype=np.int32), np.asarray(vals, dtype=np.float64)],
            "shape": tuple(int(s) for s in shape)}


# ---------------------------------------------------------------- equality_init (:39-52) and its Jacobian (:55-115)
def equality_init(x, sp):
    m, r, v, q, _, _ = split(x, sp["M"], sp["N"])
    u = sp["units"]
    parts = [] if sp["payload_mode"] else [m[0] - sp["init"]["mass"] / u["mass"]]
    parts += [r[0] - sp["init"]["position"] / u["position"], v[0] - sp["init"]["velocity"] / u["velocity"],
              q[0] - sp["init"]["quaternion"]]
    return np.concatenate(parts, axis=None)


def equality_jac_init(x, sp):
    M = sp["M"]
    o = 0 if sp["payload_mode"] else 1
    nrow = 10 + o
    jac = {}
    if o:
        jac["mass"] = _coo([0], [0], [1.0], (nrow, M))
    jac["position"] = _coo(range(o, o + 3), range(3), np.ones(3), (nrow, 3 * M))
    jac["velocity"] = _coo(range(o + 3, o + 6), range(3), np.ones(3), (nrow, 3 * M))
    jac["quaternion"] = _coo(range(o + 6, o + 10), range(4), np.ones(4), (nrow, 4 * M))
    return jac


# ---------------------------------------------------------------- equality_time (:118-141), Jacobian (:144-171)
def equality_time(x, sp):
    t = split(x, sp["M"], sp["N"])[5]
    ut = sp["units"]["t"]
    con = [t[0] - sp["time"][0] / ut]
    for i in range(1, sp["S"] + 1):
        k = sp["time_ref"][i]
        if k >= 0:
            con.append(t[i] - t[k] - (sp["time"][i] - sp["time"][k]) / ut)
    return np.concatenate(con, axis=None)


def equality_jac_time(x, sp):
    rows, cols, vals = [0], [0], [1.0]
    r = 1
    for i in range(1, sp["S"] + 1):
        k = sp["time_ref"][i]
        if k >= 0:
            rows += [r, r]; cols += [i, k]; vals += [1.0, -1.0]
            r += 1
    return {"t": _coo(rows, cols, vals, (r, sp["S"] + 1))}


# ---------------------------------------------------------------- equality_knot_LGR (:174-252), Jacobian (:255-326)
def equality_knot_LGR(x, sp):
    m, r, v, q, _, _ = split(x, sp["M"], sp["N"])
    um = sp["units"]["mass"]
    con = []
    seps = []
    for ig, sep, mass_stage in sp["stages"]:          # mass of a whole stage between its ignition and its separation
        seps.append(sep)
        con.append(m[sp["xa"][ig]] - m[sp["xa"][sep]] - mass_stage / um)
    for i in range(1, sp["S"]):
        a = sp["xa"][i]
        if i not in seps:
            con.append(m[a] - m[a - 1] + sp["mass_jettison"][i] / um)
        con += [r[a] - r[a - 1], v[a] - v[a - 1], q[a] - q[a - 1]]
    return np.concatenate(con, axis=None)


def equality_jac_knot_LGR(x, sp):
    M = sp["M"]
    E = {k: ([], [], []) for k in ("mass", "position", "velocity", "quaternion")}

    def put(key, rows, cols, val):
        E[key][0].extend(rows); E[key][1].extend(cols); E[key][2].extend([val] * len(rows))

    row = 0
    seps = []
    for ig, sep, _ in sp["stages"]:
        seps.append(sep)
        E["mass"][0].extend([row, row]); E["mass"][1].extend([sp["xa"][ig], sp["xa"][sep]]); E["mass"][2].extend([1.0, -1.0])
        row += 1
    for i in range(1, sp["S"]):
        a = sp["xa"][i]
        if i not in seps:
            E["mass"][0].extend([row, row]); E["mass"][1].extend([a - 1, a]); E["mass"][2].extend([-1.0, 1.0])
            row += 1
        for key, w in (("position", 3), ("velocity", 3), ("quaternion", 4)):
            rr = list(range(row, row + w))
            put(key, rr, range((a - 1) * w, a * w), -1.0)
            put(key, rr, range(a * w, (a + 1) * w), 1.0)
            row += w
    shapes = {"mass": M, "position": 3 * M, "velocity": 3 * M, "quaternion": 4 * M}
    return {k: _coo(E[k][0], E[k][1], E[k][2], (row, shapes[k])) for k in E}


# ---------------------------------------------------------------- orbital point functions
def angular_momentum(r, v):                # src/wrapper_coordinate.hpp:222-228
    return float(np.linalg.norm(np.cross(r, v)))


def orbit_energy(r, v):                    # :246-250
    return 0.5 * float(np.linalg.norm(v)) ** 2 - MU / float(np.linalg.norm(r))


def inclination_rad(r, v):                 # :229-236
    c = np.cross(r, v)
    return math.acos(c[2] / np.linalg.norm(c))


def angular_momentum_from_altitude(ha, hp):  # :252-258
    ra, rp = RA + ha, RA + hp
    return rp * math.sqrt(MU * (2.0 / rp - 1.0 / ((ra + rp) / 2.0)))


def orbit_energy_from_altitude(ha, hp):    # :260-265
    return -MU / 2.0 / ((RA + ha + RA + hp) / 2.0)


def orbital_elements(r, v):
    """src/Coordinate.cpp:197-245 with the degree conversion of src/wrapper_coordinate.hpp:201-209:
    (a, e, inclination, ascending node, argument of perigee, true anomaly), angles in degrees."""
    r, v = np.asarray(r, dtype=np.float64), np.asarray(v, dtype=np.float64)
    nr = r / np.linalg.norm(r)
    c = np.cross(r, v)
    f = np.cross(v, c) - MU * nr
    c1, f1 = c / np.linalg.norm(c), f / np.linalg.norm(f)
    inc = math.acos(c1[2])
    if inc > 1.0e-10:
        node = math.atan2(c1[0], -c1[1])
        argp = math.acos(math.cos(node) * f1[0] + math.sin(node) * f1[1])
        if f[2] < 0.0:
            argp = -argp
    else:
        node = 0.0
        argp = math.atan2(f[1], f[0]) if np.linalg.norm(f) > 1.0e-10 else 0.0
    e = float(np.linalg.norm(f)) / MU
    a = (float(np.dot(c, c)) / MU) / (1.0 - e * e)
    nu = math.acos(float(np.dot(f1, nr)))
    if np.dot(v, r) < 0.0:
        nu = 2.0 * math.pi - nu
    two_pi = 2.0 * math.pi
    node, argp, nu = (node + two_pi if node < 0 else node), (argp + two_pi if argp < 0 else argp), (nu + two_pi if nu < 0 else nu)
    return np.array([a, e, math.degrees(inc), math.degrees(node), math.degrees(argp), math.degrees(nu)])


# ---------------------------------------------------------------- equality_6DoF_LGR_terminal (:329-375), Jacobian (:378-405)
def _terminal_targets(sp):
    T = sp["terminal"]
    if T["altitude_perigee"] is not None and T["altitude_apogee"] is not None:
        return (angular_momentum_from_altitude(T["altitude_perigee"], T["altitude_apogee"]),
                orbit_energy_from_altitude(T["altitude_perigee"], T["altitude_apogee"]))
    c_target = T["radius"] * T["vel_tangential_geocentric"]
    vf = T["vel_tangential_geocentric"] / math.cos(math.radians(T["flightpath_vel_inertial_geocentric"]))
    return c_target, vf ** 2 / 2.0 - MU / T["radius"]


def equality_terminal(x, sp):
    _, r, v, _, _, _ = split(x, sp["M"], sp["N"])
    rf, vf = r[-1] * sp["units"]["position"], v[-1] * sp["units"]["velocity"]
    c_target, e_target = _terminal_targets(sp)
    con = [orbit_energy(rf, vf) / e_target - 1.0, angular_momentum(rf, vf) / c_target - 1.0]
    if sp["terminal"]["inclination"] is not None:
        con.append(inclination_rad(rf, vf) - math.radians(sp["terminal"]["inclination"]))
    return np.array(con)


def equality_jac_terminal(x, sp):
    M, dx = sp["M"], sp["dx"]
    fc = equality_terminal(x, sp)
    nrow = len(fc)
    jac = {}
    for key, off in (("position", M), ("velocity", 4 * M)):
        rows, cols, vals = [], [], []
        for j in range(3 * M - 3, 3 * M):
            xp = x.copy()
            xp[off + j] += dx
            fp = equality_terminal(xp, sp)
            rows += list(range(nrow)); cols += [j] * nrow; vals += ((fp - fc) / dx).tolist()
        jac[key] = _coo(rows, cols, vals, (nrow, 3 * M))
    return jac


# ---------------------------------------------------------------- inequality_time (:408-421), Jacobian (:424-452)
def _free_gaps(sp):
    return [i for i in range(sp["S"]) if not (sp["time_ref"][i] >= 0 and sp["time_ref"][i + 1] >= 0)]


def inequality_time(x, sp):
    t = split(x, sp["M"], sp["N"])[5]
    return np.array([t[i + 1] - t[i] for i in _free_gaps(sp)])


def inequality_jac_time(x, sp):
    rows, cols, vals = [], [], []
    for k, i in enumerate(_free_gaps(sp)):
        rows += [k, k]; cols += [i, i + 1]; vals += [-1.0, 1.0]
    return {"t": _coo(rows, cols, vals, (len(rows) // 2, sp["S"] + 1))}


# ---------------------------------------------------------------- lib/con_trajectory.py (stage mass, kick turn, body rates)
def inequality_mass(x, sp):                     # :33-60; Jacobian :63-103
    m = split(x, sp["M"], sp["N"])[0]
    return np.array([-m[sp["xa"][ig]] + m[sp["xa"][co]] + d / sp["units"]["mass"] for ig, co, d in sp["burns"]])


def inequality_jac_mass(x, sp):
    rows, cols, vals = [], [], []
    for k, (ig, co, _) in enumerate(sp["burns"]):
        rows += [k, k]; cols += [sp["xa"][ig], sp["xa"][co]]; vals += [-1.0, 1.0]
    return {"mass": _coo(rows, cols, vals, (len(sp["burns"]), sp["M"]))}


def _kick_sections(sp):
    return [i for i in range(sp["S"] - 1) if "kick" in sp["attitude"][i]]


def inequality_kickturn(x, sp):                 # :106-125; Jacobian :128-160
    u = split(x, sp["M"], sp["N"])[4] * sp["units"]["u"]
    parts = [-u[sp["ua"][i]:sp["ua"][i] + sp["nodes"][i], 0] for i in _kick_sections(sp)]
    return np.concatenate(parts, axis=None) if parts else np.zeros(0)


def inequality_jac_kickturn(x, sp):
    rows, cols, vals = [], [], []
    r = 0
    for i in _kick_sections(sp):
        a, n = sp["ua"][i], sp["nodes"][i]
        rows += list(range(r, r + n)); cols += list(range(2 * a, 2 * (a + n), 2)); vals += [-sp["units"]["u"]] * n
        r += n
    return {"u": _coo(rows, cols, vals, (r, 2 * sp["N"]))}


def _rate_rows(sp):
    """(column of the +1 term, column of the -1 term or -1) of every row of equality_6DoF_rate, in its order (:163-213)"""
    out = []
    for i in range(sp["S"]):
        a, n, att = sp["ua"][i], sp["nodes"][i], sp["attitude"][i]
        if att in ("hold", "vertical"):                       # both rates zero, node by node
            out += [(2 * a + k, -1) for k in range(2 * n)]
        elif att in ("kick-turn", "pitch"):                   # pitch rate constant, yaw rate zero
            out += [(2 * (a + k), 2 * a) for k in range(1, n)] + [(2 * (a + k) + 1, -1) for k in range(n)]
        elif att == "pitch-yaw":                              # both rates constant
            out += [(2 * (a + k), 2 * a) for k in range(1, n)] + [(2 * (a + k) + 1, 2 * a + 1) for k in range(1, n)]
        elif att == "same-rate":                              # both rates as at the end of the previous section
            out += [(2 * (a + k), 2 * a - 2) for k in range(n)] + [(2 * (a + k) + 1, 2 * a - 1) for k in range(n)]
        elif att not in ("zero-lift-turn", "free"):
            raise ValueError("unknown attitude option %r" % att)
    return out


def equality_rate(x, sp):
    u = split(x, sp["M"], sp["N"])[4].ravel()
    return np.array([u[p] - (u[q] if q >= 0 else 0.0) for p, q in _rate_rows(sp)])


def equality_jac_rate(x, sp):                   # :255-347: per block the -1 column entries first, then the +1 entries
    rows, cols, vals = [], [], []
    r = 0
    for i in range(sp["S"]):
        a, n, att = sp["ua"][i], sp["nodes"][i], sp["attitude"][i]

        def block(count, minus_col, plus_cols):
            nonlocal r
            rr = list(range(r, r + count))
            if minus_col is not None:
                rows.extend(rr); cols.extend([minus_col] * count); vals.extend([-1.0] * count)
            rows.extend(rr); cols.extend(plus_cols); vals.extend([1.0] * count)
            r += count

        if att in ("hold", "vertical"):
            block(2 * n, None, list(range(2 * a, 2 * (a + n))))
        elif att in ("kick-turn", "pitch"):
            block(n - 1, 2 * a, list(range(2 * (a + 1), 2 * (a + n), 2)))
            block(n, None, list(range(2 * a + 1, 2 * (a + n) + 1, 2)))
        elif att == "pitch-yaw":
            block(n - 1, 2 * a, list(range(2 * (a + 1), 2 * (a + n), 2)))
            block(n - 1, 2 * a + 1, list(range(2 * (a + 1) + 1, 2 * (a + n) + 1, 2)))
        elif att == "same-rate":
            block(n, 2 * a - 2, list(range(2 * a, 2 * (a + n), 2)))
            block(n, 2 * a - 1, list(range(2 * a + 1, 2 * (a + n) + 1, 2)))
    return {"u": _coo(rows, cols, vals, (r, 2 * sp["N"]))}


# ---------------------------------------------------------------- the shipped user constraint and lib/jac_fd.py
def user_apogee_height(x, sp, section):
    """example/user_constraints.py:120-139: elements of the first state node of `section`; a (1 - e) / 6378137 - 1."""
    _, r, v, _, _, _ = split(x, sp["M"], sp["N"])
    a = sp["xa"][section]
    el = orbital_elements(r[a] * sp["units"]["position"], v[a] * sp["units"]["velocity"])
    return (el[0] * (1.0 - el[1]) / 6378137.0) - 1.0


def jac_fd_dense(fn, x, dx):
    """lib/jac_fd.py:29-62 over the packed vector: one evaluation per column, in place += dx / -= dx on a private copy."""
    x = x.copy()
    g0 = np.atleast_1d(fn(x))
    J = np.zeros((len(g0), x.size))
    for i in range(x.size):
        x[i] += dx
        J[:, i] = (np.atleast_1d(fn(x)) - g0) / dx
        x[i] -= dx
    return J


# ---------------------------------------------------------------- initialize.py:322-409 (LGR mode)
def initial_guess(t_ref, table, knot_times, nodes, taus, units):
    """Linear interpolation (scipy interp1d, fill_value="extrapolate": slope * (t - t_lo) + y_lo on the bracketing
    interval, the end intervals extended) of the reference trajectory at the state-node and control-node times.
    table columns: mass | pos 3 | vel 3 | quat 4 | rate_y, rate_z."""
    t_ref, table = np.asarray(t_ref, dtype=np.float64), np.asarray(table, dtype=np.float64)
    tn, tx = [], []
    for i, n in enumerate(nodes):
        to, tf = knot_times[i], knot_times[i + 1]
        tau = np.asarray(taus[i])
        tn.append(tau * (tf - to) / 2.0 + (tf + to) / 2.0)
        tx.append(np.hstack((-1.0, tau)) * (tf - to) / 2.0 + (tf + to) / 2.0)
    tn, tx = np.concatenate(tn), np.concatenate(tx)

    def interp(tq, cols):
        hi = np.clip(np.searchsorted(t_ref, tq), 1, len(t_ref) - 1)      # scipy: first knot >= t, clipped
        lo = hi - 1
        slope = (table[hi][:, cols] - table[lo][:, cols]) / (t_ref[hi] - t_ref[lo])[:, None]
        return slope * (tq - t_ref[lo])[:, None] + table[lo][:, cols]

    return {"mass": (interp(tx, [0]) / units["mass"]).ravel(), "position": (interp(tx, [1, 2, 3]) / units["position"]).ravel(),
            "velocity": (interp(tx, [4, 5, 6]) / units["velocity"]).ravel(), "quaternion": interp(tx, [7, 8, 9, 10]).ravel(),
            "u": (interp(tn, [11, 12]) / units["u"]).ravel(), "t": np.asarray(knot_times, dtype=np.float64) / units["t"]}
