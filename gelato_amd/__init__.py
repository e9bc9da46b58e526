"""gelato_amd -- MI355X-native LGR defect-residual / FD-Jacobian engine behind GELATO's
pyoptsparse callback surface.

Host-side mirror of the reference interface for the hot path only:

  reference                         here
  lib/con_dynamics.py           ->  gelato_amd.con_dynamics   (same 8 functions, same signature)
  lib/dynamics_c (pybind11)     ->  gelato_amd.dynamics
  lib/PSfunctions.py (LGR part) ->  gelato_amd.PSfunctions
  lib/SectionParameters.py      ->  gelato_amd.SectionParameters
  lib/jac_fd.py                 ->  gelato_amd.jac_fd
  lib/cost_gradient.py          ->  gelato_amd.cost_gradient

All numerics run in hand-written HIP kernels (gelato_amd/csrc) through the C-ABI in
include/gelato_amd.h; there is no CPU fallback.
"""
from .engine import Engine, pack_x  # noqa: F401

__version__ = "0.1.0"
