"""Tools for user-defined constraints: the host helpers of the reference's lib/usercon_tools.py:28-162 (same names,
same arguments) and the DEVICE form of a user constraint.

A user constraint that is a function of the position and velocity at one knot -- like the shipped example
(example/user_constraints.py:120-139: orbital elements of the state at IIP_END, a (1 - e) / 6378137 - 1) -- is declared as
a `NodeFunction` row instead of Python arithmetic:

    # user_constraints.py
    from gelato_amd.usercon_tools import NodeFunction
    EQUALITY_ROWS = [NodeFunction("periapsis_radius", section="IIP_END", scale=6378137.0, offset=1.0)]

gelato_amd.con_user then evaluates value = f(r, v) / scale - offset on the device together with the knot / terminal rows,
and its Jacobian by a forward difference over the six columns the row can see, formed in the kernel (the reference's
lib/jac_fd.py:29-62 re-runs the Python function once for each of the 1,003 ... 20,113 columns and gets an exact zero for
all but those six).  A module that defines only plain ``equality_user`` / ``inequality_user`` functions keeps working: it
is called column by column exactly like the reference does.
"""
import numpy as np

from .engine import Engine


class NodeFunction(tuple):
    """(function, section, scale, offset): value = function(position, velocity at the first state node of `section`)
    / scale - offset.  Functions: Engine.NODE_FUNCTIONS (orbit_energy, angular_momentum, inclination_rad,
    semi_major_axis, eccentricity, periapsis_radius, apoapsis_radius, radius, speed)."""

    def __new__(cls, function, section, scale=1.0, offset=0.0):
        if function not in Engine.NODE_FUNCTIONS:
            raise ValueError("unknown node function %r; expected one of %s" % (function, sorted(Engine.NODE_FUNCTIONS)))
        if not scale:
            raise ValueError("scale must be non-zero")
        return super().__new__(cls, (function, section, float(scale), float(offset)))


def get_index_event(pdict, section_name, key):
    """lib/usercon_tools.py:28-72: index range of `key` belonging to the section, inside xdict[key]."""
    i = pdict["event_index"][section_name]
    if key == "t":
        return i, i + 1
    ua, ub, xa, xb, _ = pdict["ps_params"].get_index(i)
    if key == "u":
        return ua * 2, ub * 2
    width = {"position": 3, "velocity": 3, "mass": 1, "quaternion": 4}.get(key)
    if width is None:
        raise ValueError(f"Unsupported key {key!r} in get_index_event; expected one of "
                         "'mass', 'position', 'velocity', 'quaternion', 'u', or 't'.")
    return xa * width, xb * width


def get_value(xdict, pdict, unitdict, section_name, key):
    """lib/usercon_tools.py:75-104: the variable at the knot that opens the section, in physical units."""
    a, _ = get_index_event(pdict, section_name, key)
    width = {"t": 0, "mass": 0, "quaternion": 4, "u": 2}.get(key, 3)
    unit = unitdict.get(key, 1.0) if key in ("mass", "quaternion") else unitdict[key]
    return xdict[key][a] * unit if width == 0 else xdict[key][a:a + width] * unit


def get_values_section(xdict, pdict, unitdict, section_name, key):
    """lib/usercon_tools.py:107-162: the variable at every node of the section (state keys include the opening knot)."""
    i = pdict["event_index"][section_name]
    ps = pdict["ps_params"]
    n = ps.nodes(i)
    if key == "t":
        t = xdict[key] * unitdict[key]
        return ps.time_nodes(i, t[i], t[i + 1])
    ua, ub, xa, xb, _ = ps.get_index(i)
    if key == "mass":
        return (xdict[key] * unitdict[key])[xa:xa + n + 1]
    if key == "quaternion":
        return xdict[key].reshape(-1, 4)[xa:xa + n + 1]
    if key == "u":
        return (xdict[key].reshape(-1, 2) * unitdict[key])[ua:ua + n]
    return (xdict[key].reshape(-1, 3) * unitdict[key])[xa:xa + n + 1]
