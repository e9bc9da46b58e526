"""User-defined constraints: the reference's lib/con_user.py (the only caller of jac_fd).

The reference imports ``equality_user`` / ``inequality_user`` from a ``user_constraints`` module found on the import
path of the run (lib/con_user.py:28; ``_user_constraints_empty.py`` is the template that returns None).  Here the module
is looked up when first needed, and a run without one behaves like the empty template.

Two forms of a user module are understood:

* DEVICE form: the module lists ``EQUALITY_ROWS`` / ``INEQUALITY_ROWS`` of ``gelato_amd.usercon_tools.NodeFunction``
  (gelato_amd/examples/user_constraints.py is the shipped example/user_constraints.py:120-139 in that form).  Values and
  Jacobians come from the device row table of the handle: one launch evaluates them with the knot / terminal rows, and
  the forward difference perturbs -- inside the kernel -- only the six columns a row can see.  The Jacobian is returned
  as lib/jac_fd.py:29-62 returns it: a dense ``{key: [nRows, xdict[key].size]}`` for every key of xdict.
* CALLABLE form: plain ``equality_user`` / ``inequality_user`` Python functions.  They are the user's code, which the
  engine cannot run on the GPU: they are called column by column like the reference does (gelato_amd.jac_fd).
"""
import importlib

import numpy as np

from . import con_init_terminal_knot as _rows
from .jac_fd import jac_fd

_mod = None


def set_user_module(module):
    """Use `module` (an imported module or None for a fresh look-up of ``user_constraints``) for the user constraints."""
    global _mod
    _mod = module


def _user():
    global _mod
    if _mod is None:
        try:
            _mod = importlib.import_module("user_constraints")
        except ModuleNotFoundError:
            class _Empty:  # _user_constraints_empty.py:28-34
                @staticmethod
                def equality_user(xdict, pdict, unitdict, condition):
                    return None

                @staticmethod
                def inequality_user(xdict, pdict, unitdict, condition):
                    return None
            _mod = _Empty
    return _mod


def _device_rows(pdict):
    """(equality rows, inequality rows) of a device-form module, registered on the handle's row table"""
    m = _user()
    eq = list(getattr(m, "EQUALITY_ROWS", None) or ())
    ineq = list(getattr(m, "INEQUALITY_ROWS", None) or ())
    last = pdict.get("gelato_amd_user_rows_seen")      # (eq, ineq) as they were when the rows were last registered
    if last is not None and last[0] == eq and last[1] == ineq and "gelato_amd_user_rows" in pdict:
        return eq, ineq
    pdict["gelato_amd_user_rows_seen"] = ([tuple(r) for r in eq], [tuple(r) for r in ineq])   # (rows given as lists: compared in full every time)
    rows = tuple(tuple(r) for r in eq + ineq)
    if "gelato_amd_user_rows" not in pdict or tuple(tuple(r) for r in (pdict["gelato_amd_user_rows"] or ())) != rows:
        pdict["gelato_amd_user_rows"] = rows
    return eq, ineq


def _device_eval(xdict, pdict, unitdict, condition, which, need_jac=False):
    eq, ineq = _device_rows(pdict)
    mine = eq if which == 0 else ineq
    if not mine:
        return None, None, None
    R = _rows.rows_of(pdict, unitdict, condition)
    con, jfn = R.evaluate(xdict, pdict, need_jac)
    first = R.nlin + R.n_terminal + (0 if which == 0 else len(eq))
    sl = slice(first, first + len(mine))
    nodes = R.user_nodes[(0 if which == 0 else len(eq)):][:len(mine)]
    return con[sl], (jfn[sl.start - R.nlin:sl.stop - R.nlin] if need_jac else None), nodes


def _device_form(m, which):
    """None: the module gives a plain function for this kind; else its (possibly empty) list of device rows"""
    attr = ("EQUALITY_ROWS", "INEQUALITY_ROWS")[which]
    return (getattr(m, attr) or []) if hasattr(m, attr) else None


def _values(xdict, pdict, unitdict, condition, which, name):
    m = _user()
    rows = _device_form(m, which)
    if rows is None:
        return getattr(m, name)(xdict, pdict, unitdict, condition)
    if not rows:
        return None                                                      # like a function that returns None
    con, _, _ = _device_eval(xdict, pdict, unitdict, condition, which)
    return np.float64(con[0]) if len(con) == 1 else con.copy()          # the shipped example returns a scalar


def _jacobian(xdict, pdict, unitdict, condition, which, name):
    m = _user()
    rows = _device_form(m, which)
    if rows is not None and not rows:
        return None
    if rows:
        con, jfn, nodes = _device_eval(xdict, pdict, unitdict, condition, which, need_jac=True)
        jac = {key: np.zeros((len(con), np.asarray(val).size)) for key, val in xdict.items()}   # lib/jac_fd.py:54-55
        for r, node in enumerate(nodes):
            jac["position"][r, 3 * node:3 * node + 3] = jfn[r, 0:3]
            jac["velocity"][r, 3 * node:3 * node + 3] = jfn[r, 3:6]
        return jac
    f = getattr(m, name)
    if f(xdict, pdict, unitdict, condition) is not None:
        return jac_fd(f, xdict, pdict, unitdict, condition)
    return None


def equality_user(xdict, pdict, unitdict, condition):
    return _values(xdict, pdict, unitdict, condition, 0, "equality_user")


def inequality_user(xdict, pdict, unitdict, condition):
    return _values(xdict, pdict, unitdict, condition, 1, "inequality_user")


def equality_jac_user(xdict, pdict, unitdict, condition):
    """Jacobian of user-defined equality constraint."""
    return _jacobian(xdict, pdict, unitdict, condition, 0, "equality_user")


def inequality_jac_user(xdict, pdict, unitdict, condition):
    """Jacobian of user-defined inequality constraint."""
    return _jacobian(xdict, pdict, unitdict, condition, 1, "inequality_user")
