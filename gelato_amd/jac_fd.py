"""Generic forward-difference Jacobian: interface of the reference's lib/jac_fd.py:29-62.

``jac_fd(con, xdict, pdict, unitdict, condition)`` returns ``{key: dense ndarray [nRows, xdict[key].size]}``
with column i = (con(x + dx e_i) - con(x)) / dx, for *every* key of xdict (also those a constraint does not
depend on, lib/jac_fd.py:54-60).

* ``con`` is one of the four defect residual functions of gelato_amd.con_dynamics: all num_vars + 1 perturbed
  decision vectors are built and evaluated in ONE batched launch on the device (one residual evaluation per
  column), and the quotient is formed by a transpose kernel (gel_jac_fd).
* ``con`` is anything else -- in the reference that is the user's own ``equality_user`` / ``inequality_user``
  from ``user_constraints.py`` (lib/con_user.py:33-42): a Python function this package knows nothing about.
  It is called once per column exactly as the reference does; nothing of the engine runs in that loop (the
  engine has no CPU path, and the user's Python cannot run on the GPU).  Unlike the reference the caller's
  arrays are never modified: the perturbation is applied to a private copy, so no += / -= round-off drifts
  into later columns.
"""
import numpy as np

from . import con_dynamics
from .engine import GROUPS, XKEYS, pack_x


def _jac_fd_user_callable(con, xdict, pdict, unitdict, condition):
    dx = pdict["dx"]
    g_base = con(xdict, pdict, unitdict, condition)
    n_rows = len(g_base) if hasattr(g_base, "__len__") else 1          # lib/jac_fd.py:49-53
    jac = {}
    for key, val in xdict.items():
        jac[key] = np.zeros((n_rows, val.size))
        work = np.array(val, dtype=np.float64, copy=True)
        xp = dict(xdict)
        xp[key] = work
        for i in range(val.size):
            work[i] = val[i] + dx
            jac[key][:, i] = (con(xp, pdict, unitdict, condition) - g_base) / dx
            work[i] = val[i]
    return jac


def jac_fd(con, xdict, pdict, unitdict, condition):
    group = con_dynamics.RESIDUAL_FUNCTIONS.get(con)
    if group is None:
        return _jac_fd_user_callable(con, xdict, pdict, unitdict, condition)
    eng = con_dynamics.engine_of(pdict, unitdict)
    # a phase's rows see only the phase's own columns: the quotients cross PCIe as per-phase blocks (a sixth of the dense matrix at
    # 6 x 64) and are laid into the zero matrices here
    blocks, rc = eng.jac_fd_blocks(group, pack_x(xdict))
    con_dynamics.note_status(pdict, rc)
    J = np.zeros((eng.nrows[GROUPS.index(group)], eng.nvars))
    for row0, cols, blk in blocks:
        # the local columns of one variable are consecutive globally: six slices per phase
        edges = np.flatnonzero(np.diff(cols) != 1) + 1
        for a, b in zip(np.concatenate([[0], edges]), np.concatenate([edges, [len(cols)]])):
            J[row0:row0 + blk.shape[0], cols[a]:cols[a] + (b - a)] = blk[:, a:b]
    cols = eng.split_x(range(eng.nvars))
    return {k: J[:, cols[k].start:cols[k].stop] for k in XKEYS if k in xdict}
