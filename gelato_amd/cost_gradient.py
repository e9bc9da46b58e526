"""Objective and its gradient (lib/cost_gradient.py:29-47 of the reference): host-side, O(1)."""
import numpy as np


def cost_6DoF(xdict, condition):
    if condition["OptimizationMode"] == "Payload":
        return -xdict["mass"][0]
    return xdict["t"][-1]


def cost_jac(xdict, condition):
    payload = condition["OptimizationMode"] == "Payload"
    key, pos, val = ("mass", 0, -1.0) if payload else ("t", -1, 1.0)
    grad = np.zeros(xdict[key].size)
    grad[pos] = val
    return {key: grad}
