"""Node-batched 3-DoF RHS on the GPU.  Same names, argument order and meaning as the reference's
pybind11 module `dynamics_c` (src/pybind_dynamics.cpp:30-114; twin lib/dynamics.py:48-120)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

_dp = C.POINTER(C.c_double)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _d(a):
    return a.ctypes.data_as(_dp)


def dynamics_velocity(mass_e, pos_eci_e, vel_eci_e, quat_eci2body, t, param, wind_table, CA_table, units, barC20=0.0):
    """acc[n,3] / unit_vel with thrust, axial aero (US-1976 + wind + CA(Mach)) and J2 gravity."""
    mass_e, pos, vel, quat, t = _f(mass_e), _f(pos_eci_e), _f(vel_eci_e), _f(quat_eci2body), _f(t)
    param, wind, ca, units = _f(param), _f(wind_table), _f(CA_table), _f(units)
    n = mass_e.shape[0]
    if pos.shape != (n, 3) or vel.shape != (n, 3) or quat.shape != (n, 4) or t.shape != (n,):
        raise TypeError("dynamics_velocity(): incompatible function arguments (shape mismatch)")
    out = np.zeros((n, 3))
    check(lib().gel_dynamics_velocity(n, _d(mass_e), _d(pos), _d(vel), _d(quat), _d(t), _d(param), _d(wind),
                                      wind.shape[0], _d(ca), ca.shape[0], _d(units), float(barC20), _d(out)))
    return out


def dynamics_velocity_NoAir(mass_e, pos_eci_e, quat_eci2body, param, units, barC20=0.0):
    mass_e, pos, quat, param, units = _f(mass_e), _f(pos_eci_e), _f(quat_eci2body), _f(param), _f(units)
    n = mass_e.shape[0]
    if pos.shape != (n, 3) or quat.shape != (n, 4):
        raise TypeError("dynamics_velocity_NoAir(): incompatible function arguments (shape mismatch)")
    out = np.zeros((n, 3))
    check(lib().gel_dynamics_velocity_NoAir(n, _d(mass_e), _d(pos), _d(quat), _d(param), _d(units), float(barC20),
                                            _d(out)))
    return out


def dynamics_quaternion(quat_eci2body, u_e, unit_u):
    quat, u = _f(quat_eci2body), _f(u_e)
    n = quat.shape[0]
    if quat.shape != (n, 4) or u.shape != (n, 2):
        raise TypeError("dynamics_quaternion(): incompatible function arguments (shape mismatch)")
    out = np.zeros((n, 4))
    check(lib().gel_dynamics_quaternion(n, _d(quat), _d(u), float(unit_u), _d(out)))
    return out


def point_eval(kind, x, aux=None):
    """Device point functions (include/gelato_amd.h: gel_point_eval)."""
    nin = [1, 3, 3, 6, 7, 1, 1, 2, 2, 8, 8, 8, 7, 4, 1][kind]
    nout = [5, 3, 3, 3, 3, 3, 1, 4, 6, 8, 16, 4, 3, 7, 2][kind]
    x = _f(x).reshape(-1, nin)
    n = x.shape[0]
    out = np.zeros((n, nout))
    a = _f(aux) if aux is not None else None
    rows = 0 if a is None else (a.shape[0] if a.ndim == 2 else 0)
    check(lib().gel_point_eval(kind, n, _d(x), _d(a) if a is not None else None, rows, _d(out)))
    return out
