#!/usr/bin/env python3
"""Exact-arithmetic known answers for the velocity-defect Jacobian (tests/golden/g15_exact_fd.npz).

Runs in the build container (needs mpmath; NOT the reference checkout): oracle/exact_fd.py evaluates the reference's RHS
formulas (src/pybind_dynamics.cpp:30-71 and what it calls, cited there line by line) in 40-digit arithmetic on exactly the fp64
inputs the reference's sweeps form (`x += dx`, then `* unit`), and differences them: the reference's finite-difference quotients
without their rounding noise.  States: the shipped example (two aerodynamic phases) and the synthetic extremes of
tests/states.py -- dense air at 55 .. 89.9 degrees latitude, every atmosphere layer from -300 m to 700 km, random positions all
over the sphere at hypersonic speed in thick air, long phases.

For every aerodynamic phase: f_c [n, 3], and -(f_p - f_c)/dx (tf - to) unit_t / 2 for the mass, position, velocity and
quaternion sweeps ([n, 3], [n, 3, 3], [n, 3, 3], [n, 3, 4]; last index = perturbed component).

Usage:  python tests/golden/make_exact_fd.py [--baseline]      (--baseline: the BASELINE.json workloads -> g15b_exact_fd_baseline.npz)"""
import os
import sys
import time

import numpy as np
from mpmath import mpf

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import oracle  # noqa: E402
import states  # noqa: E402
from oracle import exact_fd  # noqa: E402


def example_state():
    from gelato_amd import con_dynamics, pack_x, problem
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    return dict(con_dynamics.problem_arrays(pdict, unitdict)), pack_x(xdict)


def workload_state(name):
    from gelato_amd import con_dynamics, pack_x, problem
    pdict, unitdict, condition, xdict = problem.make_problem(name)
    return dict(con_dynamics.problem_arrays(pdict, unitdict)), pack_x(xdict)


STATES = {"example": example_state, "ragged": states.ragged_state, "polar": states.polar_dense_state,
          "layers": states.all_layers_state, "long": lambda: states.long_state((87, 129, 64)), "breaks": states.layer_break_state}
# the BASELINE.json workloads themselves (g15b_exact_fd_baseline.npz): every aerodynamic phase of mixed-6x64, and the two
# aerodynamic phases of stress-12x128 with the densest air (1: KICKTURN) and the highest dynamic pressure (3: ZEROLIFT_END)
BASELINE_STATES = {"mixed-6x64": lambda: workload_state("mixed-6x64"), "stress-12x128": lambda: workload_state("stress-12x128")}
BASELINE_PHASES = {"mixed-6x64": None, "stress-12x128": [1, 3]}


def main():
    out = {}
    baseline = "--baseline" in sys.argv
    for name, build in (BASELINE_STATES if baseline else STATES).items():
        prob, x = build()
        P = oracle.Problem(prob)
        prob = dict(prob)
        prob["tau"] = [P.tau(i) for i in range(P.S)]     # the oracle's own LGR nodes: what the tests hand to both sides
        out[name + "_x"] = x
        phases = [i for i in range(P.S) if prob["reference_area"][i] != 0.0]
        if name == "example":
            phases = phases[2:4]                          # two of its five aerodynamic phases are enough here
        if baseline and BASELINE_PHASES[name] is not None:
            phases = [p for p in BASELINE_PHASES[name] if p in phases]
        out[name + "_phases"] = np.array(phases, dtype=np.int32)
        for ph in phases:
            t0 = time.time()
            T = exact_fd.velocity_fd_truth(prob, x, ph, mpf(oracle.BARC20_CPP), with_alt_sensitivity=False)
            for key in ("fc", "mass", "position", "velocity", "quaternion"):
                out["%s_p%d_%s" % (name, ph, key)] = T[key]
            print("%s phase %d: %d nodes, %.1f s, max |vel/position| %.3g" % (name, ph, len(T["fc"]), time.time() - t0,
                                                                              np.abs(T["position"]).max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "g15b_exact_fd_baseline.npz" if baseline else "g15_exact_fd.npz"), **out)


if __name__ == "__main__":
    main()
