"""LGR nodes and differentiation matrix (replaces lib/PSfunctions.py:149-168,182-208 of the
reference).  Computed by the C++ generator in csrc/gel_host.hip (Newton on P_{n-1}+P_n in extended precision +
barycentric weights) for the flipped Radau set (reverse=True, the one on GELATO's hot path); the unflipped set is its
mirror image: tau = -tau_flipped reversed, and with t = -s the Lagrange derivatives obey D[k, i] = -D_flipped[n-1-k, n-i]."""
import ctypes as C

import numpy as np

from ._lib import check, lib

_dp = C.POINTER(C.c_double)


def nodes_LGR(n, reverse=True):
    """Legendre-Gauss-Radau points including +1 (PSfunctions.py:149-168)."""
    tau = np.zeros(int(n))
    check(lib().gel_lgr_nodes(int(n), tau.ctypes.data_as(_dp)))
    return tau if reverse else -tau[::-1]


def differentiation_matrix_LGR(n, reverse=True):
    """LGR differentiation matrix, n x (n+1) (PSfunctions.py:182-208)."""
    D = np.zeros((int(n), int(n) + 1))
    check(lib().gel_lgr_diffmat(int(n), D.ctypes.data_as(_dp)))
    return D if reverse else np.ascontiguousarray(-D[::-1, ::-1])
