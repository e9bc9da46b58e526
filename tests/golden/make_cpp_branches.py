#!/usr/bin/env python3
"""Known answers for the branches that exist only in the reference's C++ (the Python twins the other goldens come from lack
them), evaluated in 40-digit arithmetic (mpmath; oracle/exact_fd.py restates the formulas with their file:line):

  gravity below the polar radius   src/gravity.cpp:45-47 clamps r to b = a (1 - f) in the radial scale factors while the direction
                                   and the Legendre terms keep the true position
  interp at the table's first row   src/wrapper_utils.hpp:68-75: `lower_bound(x) - 1` is -1 at x == xp[0], so the C++ indexes
                                   xp[-1] / yp[-1] -- out of bounds, no defined value.  The engine and the oracle implement
                                   np.interp's value yp[0] there; the neighbours of xp[0] ARE defined in both.

-> tests/golden/g16_cpp_branches.npz.  Usage: python tests/golden/make_cpp_branches.py"""
import os
import sys

import numpy as np
from mpmath import mpf

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import exact_fd  # noqa: E402

BARC20 = mpf("-0.484165371736e-3")      # src/gravity.cpp:18-19


def main():
    rng = np.random.default_rng(41)
    Rb = 6378137.0 * (1.0 - 1.0 / 298.257223563)
    d = rng.standard_normal((48, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    radii = np.concatenate([np.linspace(0.05, 0.999, 30), [0.9999999, 1.0, 1.0 + 1e-12, 1.0000001, 1.001, 1.1] * 3]) * Rb
    pos = d * radii[:, None]
    pos[5] = [0.0, 0.0, 0.4 * Rb]          # on the polar axis, under ground
    pos[6] = [0.3 * Rb, 0.0, 0.0]          # in the equatorial plane
    grav = np.array([[float(v) for v in exact_fd.gravity([exact_fd.f64(c) for c in p], BARC20)] for p in pos])
    # interp: the CA table of the example, abscissae around its first and last rows and at its inner rows
    xp = np.array([0.0, 0.5, 0.8, 1.1, 1.6, 3.0, 6.0])
    yp = np.array([0.32, 0.33, 0.45, 0.62, 0.50, 0.38, 0.33])
    xs = np.array([-1.0, -1e-300, 0.0, 5e-324, 1e-300, np.nextafter(0.5, 0), 0.5, np.nextafter(0.5, 1), 0.65, 0.8, 2.9999999999, 3.0,
                   np.nextafter(6.0, 0), 6.0, np.nextafter(6.0, 7), 7.0, 1e300])
    ys = np.array([float(exact_fd.interp(exact_fd.f64(x), [exact_fd.f64(v) for v in xp], [exact_fd.f64(v) for v in yp])) for x in xs])
    np.savez_compressed(os.path.join(HERE, "g16_cpp_branches.npz"), grav_pos=pos, grav=grav, interp_xp=xp, interp_yp=yp,
                        interp_x=xs, interp_y=ys)
    print("gravity: %d points, %d under ground; interp: %d abscissae" % (len(pos), int((np.linalg.norm(pos, axis=1) < Rb).sum()), len(xs)))


if __name__ == "__main__":
    main()
