"""A user-constraint function in the style of example/user_constraints.py (orbital quantities of the final state),
shared by make_golden.py (run through the reference's lib/jac_fd.py) and tests/test_host_cpu.py (run through
gelato_amd.jac_fd).  Our own function: test data, not reference code."""
import numpy as np


def user_con(xdict, pdict, unitdict, condition):
    pos = xdict["position"].reshape(-1, 3)[-1] * unitdict["position"]
    vel = xdict["velocity"].reshape(-1, 3)[-1] * unitdict["velocity"]
    mu = 3.986004418e14
    energy = vel @ vel / 2.0 - mu / np.linalg.norm(pos)
    h = np.cross(pos, vel)
    return np.array([energy / 1.0e7, np.linalg.norm(h) / 1.0e10, xdict["mass"][-1] - xdict["t"][-1]])


def scalar_con(xdict, pdict, unitdict, condition):
    return float(xdict["quaternion"][-4:] @ xdict["quaternion"][-4:] - 1.0)
