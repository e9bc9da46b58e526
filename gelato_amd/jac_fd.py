"""Generic forward-difference Jacobian, column-batched on the GPU.

Interface of the reference's lib/jac_fd.py:29-62: ``jac_fd(con, xdict, pdict, unitdict, condition)``
returns ``{key: dense ndarray [nRows, xdict[key].size]}`` with column i = (con(x + dx e_i) - con(x))/dx,
for *every* key of xdict (also those a constraint does not depend on, lib/jac_fd.py:54-60).

Here ``con`` must be one of the four defect residual functions of gelato_amd.con_dynamics: all
num_vars + 1 perturbed decision vectors are built and evaluated in one batched launch on the device
(one residual evaluation per decision-vector column), and the quotient is formed by a transpose
kernel.  An arbitrary Python callable cannot run on the GPU and there is no CPU fallback in this
package, so anything else raises TypeError (the reference uses jac_fd only for user-defined
constraints, lib/con_user.py:33-42, which are outside the hot path).
"""
from . import con_dynamics
from .engine import XKEYS, pack_x


def jac_fd(con, xdict, pdict, unitdict, condition):
    group = con_dynamics.RESIDUAL_FUNCTIONS.get(con)
    if group is None:
        raise TypeError("gelato_amd.jac_fd runs on the device and only accepts the four "
                        "gelato_amd.con_dynamics.equality_dynamics_* functions")
    eng = con_dynamics.engine_of(pdict, unitdict)
    J, rc = eng.jac_fd(group, pack_x(xdict))
    pdict[con_dynamics._KEY].status = rc
    cols = eng.split_x(range(eng.nvars))
    return {k: J[:, cols[k].start:cols[k].stop] for k in XKEYS if k in xdict}
