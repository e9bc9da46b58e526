"""pyoptsparse-free driver for the hot-path share of GELATO's callbacks.

`make_callbacks` returns ``objfunc(xdict) -> (funcs, fail)`` and ``sens(xdict, funcs) ->
(funcsSens, fail)`` with the reference's keys for the objective and the four defect groups
(Trajectory_Optimization.py:194-312), so a maintainer can hand them to pyoptsparse's
``Optimization(name, objfunc)`` / ``opt(optProb, sens=sens)`` exactly as the reference does
(:315,:458).  `mock_optimizer_loop` drives them the way :354-355,:458 do, counting the same
timers the reference prints (:511-517).
"""
import time

from . import con_aero, con_dynamics
from .cost_gradient import cost_6DoF, cost_jac

# wrt map of the four groups (Trajectory_Optimization.py:361-364)
WRT = {
    "eqcon_dyn_mass": ["mass", "t"],
    "eqcon_dyn_pos": ["position", "velocity", "t"],
    "eqcon_dyn_vel": ["mass", "position", "velocity", "quaternion", "t"],
    "eqcon_dyn_quat": ["quaternion", "u", "t"],
}


def make_callbacks(pdict, unitdict, condition):
    # the aero path constraints take part when the condition dict carries any of their tables
    aero = any(k in condition for k in ("AOA_max", "dynamic_pressure_max", "Q_alpha_max"))

    def objfunc(xdict):
        funcs = {"obj": cost_6DoF(xdict, condition)}
        funcs["eqcon_dyn_mass"] = con_dynamics.equality_dynamics_mass(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_pos"] = con_dynamics.equality_dynamics_position(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_vel"] = con_dynamics.equality_dynamics_velocity(xdict, pdict, unitdict, condition)
        funcs["eqcon_dyn_quat"] = con_dynamics.equality_dynamics_quaternion(xdict, pdict, unitdict, condition)
        fail = con_dynamics.last_status(pdict)
        if aero:  # Trajectory_Optimization.py:214-221 (None when a kind has no entry, like the reference)
            funcs["ineqcon_alpha"] = con_aero.inequality_max_alpha(xdict, pdict, unitdict, condition)
            funcs["ineqcon_q"] = con_aero.inequality_max_q(xdict, pdict, unitdict, condition)
            funcs["ineqcon_qalpha"] = con_aero.inequality_max_qalpha(xdict, pdict, unitdict, condition)
            fail |= con_dynamics.last_status(pdict)
        return funcs, bool(fail)

    def sens(xdict, funcs):
        fs = {"obj": cost_jac(xdict, condition)}
        fs["eqcon_dyn_mass"] = con_dynamics.equality_jac_dynamics_mass(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_pos"] = con_dynamics.equality_jac_dynamics_position(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_vel"] = con_dynamics.equality_jac_dynamics_velocity(xdict, pdict, unitdict, condition)
        fs["eqcon_dyn_quat"] = con_dynamics.equality_jac_dynamics_quaternion(xdict, pdict, unitdict, condition)
        fail = con_dynamics.last_status(pdict)
        if aero:  # Trajectory_Optimization.py:286-295
            fs["ineqcon_alpha"] = con_aero.inequality_jac_max_alpha(xdict, pdict, unitdict, condition)
            fs["ineqcon_q"] = con_aero.inequality_jac_max_q(xdict, pdict, unitdict, condition)
            fs["ineqcon_qalpha"] = con_aero.inequality_jac_max_qalpha(xdict, pdict, unitdict, condition)
            fail |= con_dynamics.last_status(pdict)
        return fs, bool(fail)

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
