"""Defect (dynamics) equality constraints and their sparse Jacobians on the GPU.

Drop-in for the reference's lib/con_dynamics.py: the same eight functions with the same
signature ``fn(xdict, pdict, unitdict, condition)`` and the same return layouts

  equality_dynamics_{mass,position,velocity,quaternion}      -> 1-D float64 ndarray
  equality_jac_dynamics_{mass,position,velocity,quaternion}  -> {var: {"coo": [rows i4, cols i4, vals f8],
                                                                       "shape": (r, c)}}

(lib/con_dynamics.py:34-63,66-113,116-152,155-213,216-289,292-496,499-533,536-632).  pyoptsparse
calls them one after another with the same xdict (Trajectory_Optimization.py:199-210,250-261);
here the first call of a group of four evaluates ALL four on the device in one fused launch and the
other three return slices of the cached result.  xdict is never mutated (the reference perturbs it
in place and restores it, con_dynamics.py:362-370).

The static problem (device-resident D matrices, tables, pattern) is created once per pdict and
cached in ``pdict["_gelato_amd"]``.  ``fail`` is reported through ``last_status(pdict)``:
non-finite output -> 1 (the reference hard-codes fail=False).
"""
import numpy as np

from .engine import BLOCKS, Engine, _d, pack_x

_KEY = "_gelato_amd"


def problem_arrays(pdict, unitdict):
    """Flatten what the hot path reads from pdict / unitdict (Trajectory_Optimization.py:116-167)."""
    S = pdict["num_sections"]
    P = pdict["params"]
    ps = pdict["ps_params"]
    return {
        "num_nodes": np.array([ps.nodes(i) for i in range(S)], dtype=np.int32),
        "thrust": np.array([P[i]["thrust"] for i in range(S)], dtype=np.float64),
        "massflow": np.array([P[i]["massflow"] for i in range(S)], dtype=np.float64),
        "reference_area": np.array([P[i]["reference_area"] for i in range(S)], dtype=np.float64),
        "nozzle_area": np.array([P[i]["nozzle_area"] for i in range(S)], dtype=np.float64),
        "engine_on": np.array([1 if P[i]["engineOn"] else 0 for i in range(S)], dtype=np.int32),
        "attitude_hold": np.array([1 if P[i]["attitude"] in ["hold", "vertical"] else 0 for i in range(S)],
                                  dtype=np.int32),
        "units": np.array([unitdict[k] for k in ["mass", "position", "velocity", "u", "t"]], dtype=np.float64),
        "dx": float(pdict["dx"]),
        "wind_table": np.asarray(pdict["wind_table"], dtype=np.float64),
        "ca_table": np.asarray(pdict["ca_table"], dtype=np.float64),
    }


class _State:
    def __init__(self, pdict, unitdict):
        prob = problem_arrays(pdict, unitdict)
        ps = pdict["ps_params"]
        S = pdict["num_sections"]
        # D and tau are inputs of the path: whatever PSparams the caller put in pdict is used as is
        self.engine = Engine(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)],
                             barC20=float(pdict.get("barC20", 0.0)), device=int(pdict.get("device", 0)))
        self.status = 0
        self._frame = None      # everything the device produced for the last xdict (one round trip)
        self._frame_sig = None
        # begin_callback() .. end_callback(): the caller vouches that this xdict object is not modified in between, so its
        # packed copy is formed once and the frame is recognised by identity instead of an element-wise comparison
        self._pinned = None
        self._pinned_x = None
        self._pinned_cond = None
        self._pinned_user = None
        self._pinned_aero = None
        self._pinned_fr = None
        self._jd = None
        self._rd = None
        # two persistent packed-x buffers (and their ctypes pointers), used in turn: the one that is not the cached frame's receives
        # the next decision vector, so the frame's x stays intact for the comparison that decides whether the frame can be reused
        # (the handle's own pinned buffers where there is a device: the kernel then reads the vector in place)
        if int(pdict.get("device", 0)) >= 0:
            self._xb, self._xp = self.engine.pinned_x()
        else:
            self._xb = [np.empty(self.engine.nvars), np.empty(self.engine.nvars)]
            self._xp = [_d(b) for b in self._xb]

    def frame(self, xdict, need_jac):
        """All device outputs for `xdict`: the first function of a callback that asks evaluates the four defect groups,
        the knot / terminal / user row table and the aero path constraints -- whatever is configured on the handle -- in
        ONE round trip (gel_eval_callback); the other functions of the callback read their share.  A derivative asked
        for after a values-only frame of the same xdict re-evaluates with derivatives."""
        fr = self._pinned_fr
        if fr is not None and xdict is self._pinned and (fr["jac"] or not need_jac) and self._frame_sig == self.engine._cfg_gen:
            return fr                       # the callback's own frame: its status is in already
        fr = self._frame
        if xdict is self._pinned and self._pinned_x is not None:
            x = self._pinned_x
        else:
            which = 1 if (fr is not None and fr["x"] is self._xb[0]) else 0
            x = pack_x(xdict, out=self._xb[which])
            if xdict is self._pinned:
                self._pinned_x = x
        eng = self.engine
        sig = eng._cfg_gen      # a frame is only valid for the row table / aero specs it was evaluated with
        if (fr is None or self._frame_sig != sig or (need_jac and not fr["jac"])
                or not (fr["x"] is x or (fr["x"][0] == x[0] and fr["x"][-1] == x[-1] and np.array_equal(fr["x"], x)))):
            fr = dict(eng.eval_callback(x, need_jac, xptr=self._xp[0] if x is self._xb[0] else self._xp[1]))
            fr["x"], fr["jac"] = x, bool(need_jac)
            self._frame, self._frame_sig = fr, sig
        elif fr["x"] is not x and xdict is self._pinned:
            self._pinned_x = fr["x"]      # equal content: keep handing out the frame's own buffer for this callback
        self.status |= fr["rc"]             # also when the cached frame is handed out again
        if xdict is self._pinned:
            self._pinned_fr = fr
        return fr

    def residuals(self, xdict):
        res = self.frame(xdict, False)["res"]
        # views of the engine's residual vector (rewritten in place by every evaluation): built once per array
        rd = self._rd
        if rd is None or rd[0] is not res:
            rd = self._rd = (res, self.engine.split_res(res))
        return rd[1]

    def jacobians(self, xdict):
        vals = self.frame(xdict, True)["vals"]
        # the block dicts hold VIEWS of the engine's value array, which every evaluation rewrites in place: built once per array
        if self._jd is None or self._jd[0] is not vals:
            self._jd = (vals, self.engine.jac_dicts(vals))
        return self._jd[1]


def _state(pdict, unitdict):
    st = pdict.get(_KEY)
    if st is None:
        st = _State(pdict, unitdict)
        pdict[_KEY] = st
    return st


def engine_of(pdict, unitdict):
    return _state(pdict, unitdict).engine


def last_status(pdict):
    """Status of the evaluations since the last reset_status(): 0 fine, 1 some output was NaN / Inf.  It is STICKY: every
    device evaluation of this pdict (defect groups, aero path constraints, knot / terminal / user rows) ORs its status
    in, so it can be read once at the end of objfunc / sens whatever the order of the calls."""
    st = pdict.get(_KEY)
    return 0 if st is None else st.status


def reset_status(pdict):
    st = pdict.get(_KEY)
    if st is not None:
        st.status = 0


def begin_callback(pdict, xdict):
    """First line of objfunc / sens: resets the sticky status and pins `xdict` -- until end_callback() the caller vouches
    that this dict and its arrays are not modified, so the ~15 constraint functions of the callback share one packed copy
    and recognise their frame by identity (saves ~0.1 ms of host time per callback at 6 x 64)."""
    st = pdict.get(_KEY)
    if st is not None:
        st.status = 0
        st._pinned, st._pinned_x, st._pinned_cond = xdict, None, None
        st._pinned_user = st._pinned_aero = st._pinned_fr = None
        # the user module's device rows are registered BEFORE the first function of the callback asks for the row table
        # (equality_init comes before equality_user in objfunc): a table pinned without them would hand equality_user the
        # rows of another group
        from . import con_user
        con_user._device_rows(pdict)


def end_callback(pdict):
    """Last line of objfunc / sens: -> the callback's status (0 fine, 1 some output was NaN / Inf); unpins xdict."""
    st = pdict.get(_KEY)
    if st is None:
        return 0
    st._pinned, st._pinned_x, st._pinned_cond = None, None, None
    st._pinned_user = st._pinned_aero = st._pinned_fr = None
    return st.status


def note_status(pdict, rc):
    st = pdict.get(_KEY)
    if st is not None:
        st.status |= int(rc)


def _copy_jac(j, pdict=None):
    # values are copied so that the caller may keep them across calls, like the reference's fresh arrays.
    # pdict["gelato_amd_share_values"] = True hands out the engine's own value arrays instead (updated in place
    # by the next evaluation): pyoptsparse copies what it is given into its own matrices straight away, and the
    # fresh copies of 607 k values are half of a sens() call at 6 x 64.
    if pdict is not None and pdict.get("gelato_amd_share_values"):
        return dict(j)      # the block dicts themselves (views of the engine's value array, built once per array)
    return {var: {"coo": [blk["coo"][0], blk["coo"][1], blk["coo"][2].copy()], "shape": blk["shape"]}
            for var, blk in j.items()}


def equality_dynamics_mass(xdict, pdict, unitdict, condition):
    """Equality constraint about dynamics of mass."""
    return _state(pdict, unitdict).residuals(xdict)["mass"].copy()


def equality_jac_dynamics_mass(xdict, pdict, unitdict, condition):
    """Jacobian of equality_dynamics_mass."""
    return _copy_jac(_state(pdict, unitdict).jacobians(xdict)["mass"], pdict)


def equality_dynamics_position(xdict, pdict, unitdict, condition):
    """Equality constraint about dynamics of position."""
    return _state(pdict, unitdict).residuals(xdict)["pos"].copy()


def equality_jac_dynamics_position(xdict, pdict, unitdict, condition):
    """Jacobian of equality_dynamics_position."""
    return _copy_jac(_state(pdict, unitdict).jacobians(xdict)["pos"], pdict)


def equality_dynamics_velocity(xdict, pdict, unitdict, condition):
    """Equality constraint about dynamics of velocity."""
    return _state(pdict, unitdict).residuals(xdict)["vel"].copy()


def equality_jac_dynamics_velocity(xdict, pdict, unitdict, condition):
    """Jacobian of equality_dynamics_velocity."""
    return _copy_jac(_state(pdict, unitdict).jacobians(xdict)["vel"], pdict)


def equality_dynamics_quaternion(xdict, pdict, unitdict, condition):
    """Equality constraint about dynamics of quaternion."""
    return _state(pdict, unitdict).residuals(xdict)["quat"].copy()


def equality_jac_dynamics_quaternion(xdict, pdict, unitdict, condition):
    """Jacobian of equality_dynamics_quaternion."""
    return _copy_jac(_state(pdict, unitdict).jacobians(xdict)["quat"], pdict)


RESIDUAL_FUNCTIONS = {
    equality_dynamics_mass: "mass",
    equality_dynamics_position: "pos",
    equality_dynamics_velocity: "vel",
    equality_dynamics_quaternion: "quat",
}
__all__ = ["equality_dynamics_mass", "equality_jac_dynamics_mass", "equality_dynamics_position",
           "equality_jac_dynamics_position", "equality_dynamics_velocity", "equality_jac_dynamics_velocity",
           "equality_dynamics_quaternion", "equality_jac_dynamics_quaternion", "BLOCKS"]
