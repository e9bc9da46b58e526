"""Stage-propellant, kick-turn and body-rate constraints on the GPU: drop-in for the reference's lib/con_trajectory.py
(same seven function names, ``fn(xdict, pdict, unitdict, condition)``, same return layouts; con_trajectory.py:33-347).

All three groups are differences (or multiples) of single decision variables: they are more linear rows of the
device row table that gelato_amd.con_init_terminal_knot builds on the handle of the defect path, evaluated in the
callback's one device round trip; their Jacobians are constants laid out in the reference's emission order."""
from . import con_init_terminal_knot as _rows


def _need_stage_events(pdict, unitdict, condition):
    """the reference indexes an empty match list when a RocketStage's ignition_at / cutoff_at events are not section names
    (lib/con_trajectory.py:40-49): IndexError, from these two functions only"""
    miss = _rows.rows_of(pdict, unitdict, condition).missing_stage_events
    if miss is not None:
        raise IndexError("inequality_mass: RocketStage events %r / %r are not section names (lib/con_trajectory.py:40-49 "
                         "raises IndexError too)" % miss)


def inequality_mass(xdict, pdict, unitdict, condition):
    """Inequality constraint about the propellant a stage may burn (a list, like the reference returns)."""
    _need_stage_events(pdict, unitdict, condition)
    return list(_rows._values(xdict, pdict, unitdict, condition, "imass"))


def inequality_jac_mass(xdict, pdict, unitdict, condition):
    _need_stage_events(pdict, unitdict, condition)
    return _rows._const_jac(pdict, unitdict, condition, "imass")


def inequality_kickturn(xdict, pdict, unitdict, condition):
    """Inequality constraint about the sign of the pitch rate in kick-turn sections."""
    return _rows._values(xdict, pdict, unitdict, condition, "kick")


def inequality_jac_kickturn(xdict, pdict, unitdict, condition):
    return _rows._const_jac(pdict, unitdict, condition, "kick")


def equality_6DoF_rate(xdict, pdict, unitdict, condition):
    """Equality constraint about the body rates of every attitude option."""
    return _rows._values(xdict, pdict, unitdict, condition, "rate")


def equality_length_6DoF_rate(xdict, pdict, unitdict, condition):
    a, b = _rows.rows_of(pdict, unitdict, condition).slices["rate"]
    return b - a


def equality_jac_6DoF_rate(xdict, pdict, unitdict, condition):
    return _rows._const_jac(pdict, unitdict, condition, "rate")
