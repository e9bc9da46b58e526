"""LGR nodes and differentiation matrix (replaces lib/PSfunctions.py:149-168,182-208 of the
reference; flipped Radau points only, i.e. reverse=True).  Computed by the C++ generator in
csrc/gel_host.hip (Newton on P_{n-1}+P_n in extended precision + barycentric weights)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

_dp = C.POINTER(C.c_double)


def nodes_LGR(n, reverse=True):
    """Legendre-Gauss-Radau points including +1 (PSfunctions.py:149-168)."""
    if not reverse:
        raise NotImplementedError("only the flipped LGR set (reverse=True) is on GELATO's hot path")
    tau = np.zeros(int(n))
    check(lib().gel_lgr_nodes(int(n), tau.ctypes.data_as(_dp)))
    return tau


def differentiation_matrix_LGR(n, reverse=True):
    """LGR differentiation matrix, n x (n+1) (PSfunctions.py:182-208)."""
    if not reverse:
        raise NotImplementedError("only the flipped LGR set (reverse=True) is on GELATO's hot path")
    D = np.zeros((int(n), int(n) + 1))
    check(lib().gel_lgr_diffmat(int(n), D.ctypes.data_as(_dp)))
    return D
