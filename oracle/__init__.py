"""ctypes loader for the CPU ORACLE (test infrastructure, NOT the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; ``gelato_amd`` never does.  See
``oracle/gelato_oracle.h`` for scope, provenance (reference file:line per
function) and how parity is pinned (tests/golden/, tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GROUPS = ["mass", "pos", "vel", "quat"]
BLOCK_VARS = {
    "mass": ["mass", "t"],
    "pos": ["position", "velocity", "t"],
    "vel": ["mass", "position", "velocity", "quaternion", "t"],
    "quat": ["quaternion", "u", "t"],
}
BARC20_CPP = -0.484165371736e-3            # src/gravity.cpp:18-19 (production path)
BARC20_PY_TWIN = -1.082628e-3 / 5.0 ** 0.5  # J2 of lib/coordinate.py:473 expressed as C20-bar

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(_HERE, "libgelato_oracle.so")
    src = os.path.join(_HERE, "gelato_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libgelato_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        for name in ["orc_geopotential_altitude", "orc_air_temperature", "orc_air_pressure", "orc_air_density",
                     "orc_speed_of_sound"]:
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [C.c_double]
        L.orc_interp.restype = C.c_double
        L.orc_interp.argtypes = [C.c_double, _dp, _dp, C.c_int, C.c_int]
        L.orc_problem_create.restype = C.c_void_p
        L.orc_problem_create.argtypes = [C.c_int, _ip, _dp, _dp, _dp, _dp, _ip, _ip, _dp, C.c_double, C.c_double,
                                         _dp, C.c_int, _dp, C.c_int, _dp, _dp]
        L.orc_problem_destroy.argtypes = [C.c_void_p]
        L.orc_num_vars.argtypes = [C.c_void_p]
        L.orc_num_rows.argtypes = [C.c_void_p, C.c_int]
        L.orc_block_nnz.restype = C.c_int64
        L.orc_block_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_total_nnz.restype = C.c_int64
        L.orc_total_nnz.argtypes = [C.c_void_p]
        L.orc_block_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_problem_D.restype = _dp
        L.orc_problem_D.argtypes = [C.c_void_p, C.c_int]
        L.orc_problem_tau.restype = _dp
        L.orc_problem_tau.argtypes = [C.c_void_p, C.c_int]
        L.orc_residual.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_jacobian.argtypes = [C.c_void_p, C.c_int, _dp, _ip, _ip, _dp]
        L.orc_jac_fd.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.orc_cost.restype = C.c_double
        L.orc_cost.argtypes = [C.c_void_p, _dp, C.c_int]
        L.orc_cost_jac.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.orc_eval_batch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int]
        L.orc_lgr_nodes.argtypes = [C.c_int, _dp]
        L.orc_lgr_diffmat.argtypes = [C.c_int, _dp]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------- point functions ----------------
def geopotential_altitude(z): return lib().orc_geopotential_altitude(float(z))
def air_temperature(h): return lib().orc_air_temperature(float(h))
def air_pressure(h): return lib().orc_air_pressure(float(h))
def air_density(h): return lib().orc_air_density(float(h))
def speed_of_sound(h): return lib().orc_speed_of_sound(float(h))


def _vec_fn(name, nin, nout):
    def f(*args):
        L = lib()
        cargs = []
        for a, k in zip(args, nin):
            if k == 0:
                cargs.append(C.c_double(float(a)))
            else:
                cargs.append(_d(_f64(a)))
        out = np.zeros(nout)
        getattr(L, name)(*cargs, _d(out))
        return out
    return f


gravity = _vec_fn("orc_gravity", [3, 0], 3)              # (pos, barC20)
ecef2eci = _vec_fn("orc_ecef2eci", [3, 0], 3)
eci2ecef = _vec_fn("orc_eci2ecef", [3, 0], 3)
vel_eci2ecef = _vec_fn("orc_vel_eci2ecef", [3, 3, 0], 3)
quat_nedg2eci = _vec_fn("orc_quat_nedg2eci", [3, 0], 4)
quatmult = _vec_fn("orc_quatmult", [4, 4], 4)
quatrot = _vec_fn("orc_quatrot", [4, 3], 3)
ecef2geodetic = _vec_fn("orc_ecef2geodetic", [0, 0, 0], 3)


def interp(x, xp, yp):
    xp, yp = _f64(xp), _f64(yp)
    return lib().orc_interp(float(x), _d(xp), _d(yp), len(xp), 1)


def wind_ned(alt, wind):
    wind = _f64(wind)
    out = np.zeros(3)
    lib().orc_wind_ned(C.c_double(float(alt)), _d(wind), C.c_int(wind.shape[0]), _d(out))
    return out


def dynamics_velocity(mass_e, pos_e, vel_e, quat, t, param, wind, ca, units, barC20=BARC20_CPP):
    mass_e, pos_e, vel_e, quat, t = map(_f64, (mass_e, pos_e, vel_e, quat, t))
    param, wind, ca, units = map(_f64, (param, wind, ca, units))
    n = len(mass_e)
    out = np.zeros((n, 3))
    lib().orc_dynamics_velocity(C.c_int(n), _d(mass_e), _d(pos_e), _d(vel_e), _d(quat), _d(t), _d(param),
                                _d(wind), C.c_int(wind.shape[0]), _d(ca), C.c_int(ca.shape[0]), _d(units),
                                C.c_double(barC20), _d(out))
    return out


def dynamics_velocity_NoAir(mass_e, pos_e, quat, param, units, barC20=BARC20_CPP):
    mass_e, pos_e, quat, param, units = map(_f64, (mass_e, pos_e, quat, param, units))
    n = len(mass_e)
    out = np.zeros((n, 3))
    lib().orc_dynamics_velocity_NoAir(C.c_int(n), _d(mass_e), _d(pos_e), _d(quat), _d(param), _d(units),
                                      C.c_double(barC20), _d(out))
    return out


def dynamics_quaternion(quat, u_e, unit_u):
    quat, u_e = _f64(quat), _f64(u_e)
    n = quat.shape[0]
    out = np.zeros((n, 4))
    lib().orc_dynamics_quaternion(C.c_int(n), _d(quat), _d(u_e), C.c_double(unit_u), _d(out))
    return out


def lgr_nodes(n):
    tau = np.zeros(n)
    lib().orc_lgr_nodes(n, _d(tau))
    return tau


def lgr_diffmat(n):
    D = np.zeros((n, n + 1))
    lib().orc_lgr_diffmat(n, _d(D))
    return D


class Problem:
    """Static problem (pdict + unitdict flattened).  ``prob`` is a dict with keys
    num_nodes, thrust, massflow, reference_area, nozzle_area, engine_on,
    attitude_hold, units (mass,position,velocity,u,t), dx, wind_table, ca_table;
    optional D / tau lists (per phase) override the oracle's own LGR generator."""

    def __init__(self, prob, barC20=BARC20_CPP, D=None, tau=None):
        L = lib()
        nn = np.ascontiguousarray(prob["num_nodes"], dtype=np.int32)
        self.S = len(nn)
        self.n = nn.copy()
        self.N = int(nn.sum())
        self.M = self.N + self.S
        eo = np.ascontiguousarray(prob["engine_on"], dtype=np.int32)
        ah = np.ascontiguousarray(prob["attitude_hold"], dtype=np.int32)
        wind, ca = _f64(prob["wind_table"]), _f64(prob["ca_table"])
        units = _f64(prob["units"])
        Dall = tall = None
        if D is not None:
            Dall = _f64(np.concatenate([np.asarray(d).ravel() for d in D]))
            tall = _f64(np.concatenate([np.asarray(t).ravel() for t in tau]))
        self._h = L.orc_problem_create(
            self.S, _i(nn), _d(_f64(prob["thrust"])), _d(_f64(prob["massflow"])), _d(_f64(prob["reference_area"])),
            _d(_f64(prob["nozzle_area"])), _i(eo), _i(ah), _d(units), float(prob["dx"]), float(barC20),
            _d(wind), wind.shape[0], _d(ca), ca.shape[0],
            _d(Dall) if Dall is not None else None, _d(tall) if tall is not None else None)
        self.nvars = L.orc_num_vars(self._h)
        self.nrows = [L.orc_num_rows(self._h, g) for g in range(4)]
        self.block_nnz = {g: [int(L.orc_block_nnz(self._h, gi, b)) for b in range(len(BLOCK_VARS[g]))]
                          for gi, g in enumerate(GROUPS)}
        self.total_nnz = int(L.orc_total_nnz(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().orc_problem_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def D(self, i):
        n = int(self.n[i])
        return np.ctypeslib.as_array(lib().orc_problem_D(self._h, i), shape=(n, n + 1)).copy()

    def tau(self, i):
        n = int(self.n[i])
        return np.ctypeslib.as_array(lib().orc_problem_tau(self._h, i), shape=(n,)).copy()

    def residual(self, group, x):
        gi = GROUPS.index(group)
        x = _f64(x)
        assert x.size == self.nvars
        out = np.zeros(self.nrows[gi])
        lib().orc_residual(self._h, gi, _d(x), _d(out))
        return out

    def jacobian(self, group, x, pattern=True):
        """-> {var: {"coo": [rows i4, cols i4, vals f8], "shape": (r, c)}} like the reference."""
        gi = GROUPS.index(group)
        x = _f64(x)
        nn = self.block_nnz[group]
        tot = sum(nn)
        rows = np.zeros(tot, dtype=np.int32)
        cols = np.zeros(tot, dtype=np.int32)
        vals = np.zeros(tot)
        lib().orc_jacobian(self._h, gi, _d(x), _i(rows) if pattern else None, _i(cols) if pattern else None, _d(vals))
        out, off = {}, 0
        for b, var in enumerate(BLOCK_VARS[group]):
            sh = (C.c_int64 * 2)()
            lib().orc_block_shape(self._h, gi, b, sh)
            out[var] = {"coo": [rows[off:off + nn[b]], cols[off:off + nn[b]], vals[off:off + nn[b]]],
                        "shape": (int(sh[0]), int(sh[1]))}
            off += nn[b]
        return out

    def jac_fd(self, group, x):
        gi = GROUPS.index(group)
        x = _f64(x)
        J = np.zeros((self.nrows[gi], self.nvars))
        lib().orc_jac_fd(self._h, gi, _d(x), _d(J))
        return J

    def cost(self, x, payload_mode=True):
        return lib().orc_cost(self._h, _d(_f64(x)), int(payload_mode))

    def cost_jac(self, x, payload_mode=True):
        g = np.zeros(self.M if payload_mode else self.S + 1)
        lib().orc_cost_jac(self._h, _d(_f64(x)), int(payload_mode), _d(g))
        return g

    def eval_batch(self, X, nthreads=1, keep_vals=True):
        """keep_vals=False (timing): the Jacobian values are computed but not returned -- B rows of
        total_nnz doubles are 20 GB at B = 4096 for the 6 x 64 mesh."""
        X = _f64(X).reshape(-1, self.nvars)
        B = X.shape[0]
        res = np.zeros((B, 11 * self.N))
        vals = np.zeros((B, self.total_nnz)) if keep_vals else None
        lib().orc_eval_batch(self._h, B, _d(X), _d(res), _d(vals) if keep_vals else None, int(nthreads))
        return res, vals

    # ---- aero path constraints (lib/con_aero.py) ----
    AERO_KINDS = ["alpha", "q", "qalpha"]
    AERO_VARS = ["position", "velocity", "quaternion", "t"]

    def aero_configure(self, kind, spec):
        """spec: rows of (phase, range_all, limit) with limit = units[3] of con_aero.py."""
        L = lib()
        L.orc_aero_configure.argtypes = [C.c_void_p, C.c_int, C.c_int, _ip, _ip, _dp]
        spec = np.asarray(spec, dtype=np.float64).reshape(-1, 3)
        ph = np.ascontiguousarray(spec[:, 0], dtype=np.int32)
        ra = np.ascontiguousarray(spec[:, 1], dtype=np.int32)
        lim = _f64(spec[:, 2])
        assert L.orc_aero_configure(self._h, self.AERO_KINDS.index(kind), len(ph), _i(ph), _i(ra), _d(lim)) == 0

    def aero_residual(self, kind, x):
        L = lib()
        L.orc_aero_rows.argtypes = [C.c_void_p, C.c_int]
        L.orc_aero_residual.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        k = self.AERO_KINDS.index(kind)
        out = np.zeros(L.orc_aero_rows(self._h, k))
        L.orc_aero_residual(self._h, k, _d(_f64(x)), _d(out))
        return out

    def aero_jacobian(self, kind, x):
        L = lib()
        L.orc_aero_rows.argtypes = [C.c_void_p, C.c_int]
        L.orc_aero_nnz.restype = C.c_int64
        L.orc_aero_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_aero_jacobian.argtypes = [C.c_void_p, C.c_int, _dp, _ip, _ip, _dp]
        k = self.AERO_KINDS.index(kind)
        nrow = L.orc_aero_rows(self._h, k)
        nn = [int(L.orc_aero_nnz(self._h, k, v)) for v in range(4)]
        tot = sum(nn)
        rows, cols, vals = np.zeros(tot, np.int32), np.zeros(tot, np.int32), np.zeros(tot)
        L.orc_aero_jacobian(self._h, k, _d(_f64(x)), _i(rows), _i(cols), _d(vals))
        shapes = [(nrow, 3 * self.M), (nrow, 3 * self.M), (nrow, 4 * self.M), (nrow, self.S + 1)]
        out, off = {}, 0
        for v, var in enumerate(self.AERO_VARS):
            out[var] = {"coo": [rows[off:off + nn[v]], cols[off:off + nn[v]], vals[off:off + nn[v]]], "shape": shapes[v]}
            off += nn[v]
        return out

    def split_x(self, x):
        M, N, S = self.M, self.N, self.S
        o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N, S + 1])
        keys = ["mass", "position", "velocity", "quaternion", "u", "t"]
        return {k: x[o[i]:o[i + 1]] for i, k in enumerate(keys)}
