"""User-defined constraints: the reference's lib/con_user.py (the only caller of jac_fd).

The reference imports ``equality_user`` / ``inequality_user`` from a ``user_constraints`` module found on the
import path of the run (lib/con_user.py:28; ``_user_constraints_empty.py`` is the template that returns None).
Here the module is looked up when first needed, and a run without one behaves like the empty template.
"""
import importlib

from .jac_fd import jac_fd

_mod = None


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


def equality_user(xdict, pdict, unitdict, condition):
    return _user().equality_user(xdict, pdict, unitdict, condition)


def inequality_user(xdict, pdict, unitdict, condition):
    return _user().inequality_user(xdict, pdict, unitdict, condition)


def equality_jac_user(xdict, pdict, unitdict, condition):
    """Jacobian of user-defined equality constraint."""
    f = _user().equality_user
    if f(xdict, pdict, unitdict, condition) is not None:
        return jac_fd(f, xdict, pdict, unitdict, condition)


def inequality_jac_user(xdict, pdict, unitdict, condition):
    """Jacobian of user-defined inequality constraint."""
    f = _user().inequality_user
    if f(xdict, pdict, unitdict, condition) is not None:
        return jac_fd(f, xdict, pdict, unitdict, condition)
