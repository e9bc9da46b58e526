"""pyoptsparse-free driver for the hot-path share of GELATO's callbacks.

`make_callbacks` returns ``objfunc(xdict) -> (funcs, fail)`` and ``sens(xdict, funcs) ->
(funcsSens, fail)`` with the reference's keys for the objective and the four defect groups
(Trajectory_Optimization.py:194-312), so a maintainer can hand them to pyoptsparse's
``Optimization(name, objfunc)`` / ``opt(optProb, sens=sens)`` exactly as the reference does
(:315,:458).  `mock_optimizer_loop` drives them the way :354-355,:458 do, counting the same
timers the reference prints (:511-517).
"""
import time

from . import con_aero, con_dynamics, con_trajectory, con_user, con_waypoint
from . import con_init_terminal_knot as con_a
from .cost_gradient import cost_6DoF, cost_jac

# wrt map of every constraint group (Trajectory_Optimization.py:356-383)
WRT = {
    "eqcon_init": ["mass", "position", "velocity", "quaternion"],
    "eqcon_time": ["t"],
    "eqcon_dyn_mass": ["mass", "t"],
    "eqcon_dyn_pos": ["position", "velocity", "t"],
    "eqcon_dyn_vel": ["mass", "position", "velocity", "quaternion", "t"],
    "eqcon_dyn_quat": ["quaternion", "u", "t"],
    "eqcon_knot": ["mass", "position", "velocity", "quaternion"],
    "eqcon_terminal": ["position", "velocity"],
    "eqcon_rate": ["u"],
    "eqcon_pos": ["position", "t"],
    "eqcon_iip": ["position", "velocity", "t"],
    "eqcon_user": ["mass", "position", "velocity", "quaternion", "u", "t"],
    "ineqcon_alpha": ["position", "velocity", "quaternion", "t"],
    "ineqcon_q": ["position", "velocity", "quaternion", "t"],
    "ineqcon_qalpha": ["position", "velocity", "quaternion", "t"],
    "ineqcon_mass": ["mass"],
    "ineqcon_kick": ["u"],
    "ineqcon_time": ["t"],
    "ineqcon_pos": ["position", "t"],
    "ineqcon_iip": ["position", "velocity", "t"],
    "ineqcon_antenna": ["position", "t"],
    "ineqcon_user": ["mass", "position", "velocity", "quaternion", "u", "t"],
}


def constraint_groups(f_init, jac_init, condition):
    """What the reference hands to ``optProb.addConGroup`` for every group that exists (Trajectory_Optimization.py:385-421):
    (key, size, lower, upper, wrt, jac); equality groups are pinned to 0, inequality groups are >= 0; in Payload mode
    the initial mass is free."""
    wrt = {k: list(v) for k, v in WRT.items()}
    if condition["OptimizationMode"] == "Payload":
        wrt["eqcon_init"] = ["position", "velocity", "quaternion"]
    out = []
    for key, val in f_init.items():
        if key == "obj" or val is None:
            continue
        size = len(val) if hasattr(val, "__len__") else 1
        out.append((key, size, 0.0, None if "ineqcon" in key else 0.0, wrt[key], jac_init[key]))
    return out


def make_callbacks(pdict, unitdict, condition):
    # the aero path constraints take part when the condition dict carries any of their tables
    aero = any(k in condition for k in ("AOA_max", "dynamic_pressure_max", "Q_alpha_max"))

    # the init / time / knot / terminal rows (and user rows) take part when the problem description carries what they read
    rows = all(k in pdict for k in ("event_index", "RocketStage")) and "init" in condition
    if rows:  # a cut-down event list (the synthetic bench meshes) may lack a stage's ignition / separation events, which
        # equality_knot_LGR looks up by name (lib/con_init_terminal_knot.py:192-203): no knot rows can be formed then
        ev = pdict["event_index"]
        rows = all(st["separation_at"] is None or (st["separation_at"] in ev and st["ignition_at"] in ev)
                   for st in pdict["RocketStage"].values()) and all("attitude" in q for q in pdict["params"])

    def objfunc(xdict):
        con_dynamics.begin_callback(pdict, xdict)    # sticky status over ALL device evaluations of this callback; xdict pinned
        try:
            return _objfunc(xdict)
        except BaseException:
            con_dynamics.end_callback(pdict)         # never leave the dict pinned: a later direct con_* call must see a fresh x
            raise

    def _objfunc(xdict):
        funcs = {"obj": cost_6DoF(xdict, condition)}
        if rows:  # Trajectory_Optimization.py:197-198,212-216,220,233,241
            funcs["eqcon_init"] = con_a.equality_init(xdict, pdict, unitdict, condition)
            funcs["eqcon_time"] = con_a.equality_time(xdict, pdict, unitdict, condition)
            funcs["eqcon_knot"] = con_a.equality_knot_LGR(xdict, pdict, unitdict, condition)
            funcs["eqcon_terminal"] = con_a.equality_6DoF_LGR_terminal(xdict, pdict, unitdict, condition)
            funcs["eqcon_user"] = con_user.equality_user(xdict, pdict, unitdict, condition)
            funcs["ineqcon_time"] = con_a.inequality_time(xdict, pdict, unitdict, condition)
            funcs["ineqcon_user"] = con_user.inequality_user(xdict, pdict, unitdict, condition)
            funcs["eqcon_rate"] = con_trajectory.equality_6DoF_rate(xdict, pdict, unitdict, condition)          # :217
            funcs["ineqcon_mass"] = con_trajectory.inequality_mass(xdict, pdict, unitdict, condition)           # :228
            funcs["ineqcon_kick"] = con_trajectory.inequality_kickturn(xdict, pdict, unitdict, condition)       # :229-231
            funcs["eqcon_pos"] = con_waypoint.equality_posLLH(xdict, pdict, unitdict, condition)                # :217-218
            funcs["eqcon_iip"] = con_waypoint.equality_IIP(xdict, pdict, unitdict, condition)
            funcs["ineqcon_pos"] = con_waypoint.inequality_posLLH(xdict, pdict, unitdict, condition)            # :233-237
            funcs["ineqcon_iip"] = con_waypoint.inequality_IIP(xdict, pdict, unitdict, condition)
            funcs["ineqcon_antenna"] = con_waypoint.inequality_antenna(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_mass"] = con_dynamics.equality_dynamics_mass(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_pos"] = con_dynamics.equality_dynamics_position(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_vel"] = con_dynamics.equality_dynamics_velocity(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_quat"] = con_dynamics.equality_dynamics_quaternion(xdict, pdict, unitdict, condition)
        if aero:  # Trajectory_Optimization.py:214-221 (None when a kind has no entry, like the reference)
            funcs["ineqcon_alpha"] = con_aero.inequality_max_alpha(xdict, pdict, unitdict, condition)
            funcs["ineqcon_q"] = con_aero.inequality_max_q(xdict, pdict, unitdict, condition)
            funcs["ineqcon_qalpha"] = con_aero.inequality_max_qalpha(xdict, pdict, unitdict, condition)
        return funcs, bool(con_dynamics.end_callback(pdict))

    def sens(xdict, funcs):
        con_dynamics.begin_callback(pdict, xdict)
        try:
            return _sens(xdict, funcs)
        except BaseException:
            con_dynamics.end_callback(pdict)
            raise

    def _sens(xdict, funcs):
        fs = {"obj": cost_jac(xdict, condition)}
        if rows:  # Trajectory_Optimization.py:248-249,264-269,279-281,297-299,309-311
            fs["eqcon_init"] = con_a.equality_jac_init(xdict, pdict, unitdict, condition)
            fs["eqcon_time"] = con_a.equality_jac_time(xdict, pdict, unitdict, condition)
            fs["eqcon_knot"] = con_a.equality_jac_knot_LGR(xdict, pdict, unitdict, condition)
            fs["eqcon_terminal"] = con_a.equality_jac_6DoF_LGR_terminal(xdict, pdict, unitdict, condition)
            fs["eqcon_user"] = con_user.equality_jac_user(xdict, pdict, unitdict, condition)
            fs["ineqcon_time"] = con_a.inequality_jac_time(xdict, pdict, unitdict, condition)
            fs["ineqcon_user"] = con_user.inequality_jac_user(xdict, pdict, unitdict, condition)
            fs["eqcon_rate"] = con_trajectory.equality_jac_6DoF_rate(xdict, pdict, unitdict, condition)
            fs["ineqcon_mass"] = con_trajectory.inequality_jac_mass(xdict, pdict, unitdict, condition)
            fs["ineqcon_kick"] = con_trajectory.inequality_jac_kickturn(xdict, pdict, unitdict, condition)
            fs["eqcon_pos"] = con_waypoint.equality_jac_posLLH(xdict, pdict, unitdict, condition)               # :272-275
            fs["eqcon_iip"] = con_waypoint.equality_jac_IIP(xdict, pdict, unitdict, condition)
            fs["ineqcon_pos"] = con_waypoint.inequality_jac_posLLH(xdict, pdict, unitdict, condition)           # :298-306
            fs["ineqcon_iip"] = con_waypoint.inequality_jac_IIP(xdict, pdict, unitdict, condition)
            fs["ineqcon_antenna"] = con_waypoint.inequality_jac_antenna(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_mass"] = con_dynamics.equality_jac_dynamics_mass(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_pos"] = con_dynamics.equality_jac_dynamics_position(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_vel"] = con_dynamics.equality_jac_dynamics_velocity(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_quat"] = con_dynamics.equality_jac_dynamics_quaternion(xdict, pdict, unitdict, condition)
        if aero:  # Trajectory_Optimization.py:286-295
            fs["ineqcon_alpha"] = con_aero.inequality_jac_max_alpha(xdict, pdict, unitdict, condition)
            fs["ineqcon_q"] = con_aero.inequality_jac_max_q(xdict, pdict, unitdict, condition)
            fs["ineqcon_qalpha"] = con_aero.inequality_jac_max_qalpha(xdict, pdict, unitdict, condition)
        return fs, bool(con_dynamics.end_callback(pdict))

    return objfunc, sens


def mock_optimizer_loop(objfunc, sens, xdict, iterations=5, step=1e-7):
    """Calls objfunc/sens like pyoptsparse would (one of each per major iteration) on slightly moved
    points; returns the reference's timer names (Trajectory_Optimization.py:511-517)."""
    t0 = time.perf_counter()
    t_obj = t_sens = 0.0
    f_init, fail0 = objfunc(xdict)          # :354
    jac_init, fail1 = sens(xdict, f_init)   # :355  (fixes the sparsity pattern)
    fails = int(fail0) + int(fail1)
    x = {k: v.copy() for k, v in xdict.items()}
    for it in range(iterations):
        for k in x:
            x[k] = x[k] * (1.0 + step)
        a = time.perf_counter()
        funcs, f = objfunc(x)
        b = time.perf_counter()
        fs, g = sens(x, funcs)
        c = time.perf_counter()
        t_obj += b - a
        t_sens += c - b
        fails += int(f) + int(g)
    opt = time.perf_counter() - t0
    # the reference prints seven counters (Trajectory_Optimization.py:511-517); without an optimiser in the
    # loop optCodeTime is 0 and interfaceTime is what is left of optTime
    return {"optTime": opt, "userObjTime": t_obj, "userSensTime": t_sens, "optCodeTime": 0.0,
            "interfaceTime": max(0.0, opt - t_obj - t_sens),
            "userObjCalls": iterations, "userSensCalls": iterations, "fails": fails,
            "f_init": f_init, "jac_init": jac_init}
