"""Problem set-up for the example vehicle without pyoptsparse: restates the set-up block of the
reference driver (Trajectory_Optimization.py:55-167) and its from-file initial guess
(initialize.py:322-409) for an arbitrary choice of phases / knots / node counts, and builds the
synthetic decision-vector batches of SURVEY.md 8(d).

The vehicle data (gelato_amd/data/example_vehicle.json) are plain numbers derived from the
reference's example/ directory by tests/golden/make_golden.py.
"""
import json
import os

import numpy as np

from .SectionParameters import PSparams

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "example_vehicle.json")

# the configurations of SURVEY.md 8(d): (event rows, knot times, nodes per phase)
CONFIGS = {
    # config 2: 3 phases x 32
    "3x32": (["KICKTURN", "ZEROLIFT_START", "SEIG", "SIMEND"], [0.0, 20.0, 169.0, 597.0], [32] * 3),
    # the same three phases with 16 / 8 nodes: where the 16-row matrix-pipe tile of D.X is mostly padding
    "3x16": (["KICKTURN", "ZEROLIFT_START", "SEIG", "SIMEND"], [0.0, 20.0, 169.0, 597.0], [16] * 3),
    "3x8": (["KICKTURN", "ZEROLIFT_START", "SEIG", "SIMEND"], [0.0, 20.0, 169.0, 597.0], [8] * 3),
    # config 3: the 6-phase multi-stage vehicle, 64 nodes per phase; exercises every branch
    "mixed-6x64": (["LIFTOFF", "KICKTURN", "ZEROLIFT_START", "ZEROLIFT_END", "MECO", "SEIG", "SIMEND"],
                   [0.0, 10.0, 20.0, 90.0, 169.0, 179.0, 597.0], [64] * 6),
    # maximum work: all six phases powered, aerodynamic, free attitude
    "dense-6x64": (["ZEROLIFT_END"] * 6 + ["SIMEND"], [0.0, 10.0, 20.0, 90.0, 169.0, 179.0, 597.0], [64] * 6),
}


def load_vehicle(path=_DATA):
    with open(path) as f:
        return json.load(f)


def example_config(vehicle):
    ev = vehicle["events"]
    return [e["name"] for e in ev], [e["time"] for e in ev], [e["num_nodes"] for e in ev[:-1]]


def stress_config(vehicle, nodes=128):
    """config 5: the example's 12 phases with `nodes` LGR nodes each."""
    names, times, _ = example_config(vehicle)
    return names, times, [nodes] * (len(names) - 1)


def build_pdict(vehicle, rows, knots, nodes, ps_params=None):
    """-> (pdict, unitdict, condition).  Phase i takes the parameters of event row rows[i]."""
    wt = np.asarray(vehicle["wind_alt_speed_dir"], dtype=np.float64)
    wind_table = np.column_stack([wt[:, 0], wt[:, 1] * -np.cos(np.radians(wt[:, 2])),
                                  wt[:, 1] * -np.sin(np.radians(wt[:, 2]))])  # Trajectory_Optimization.py:55-59
    ca_table = np.asarray(vehicle["ca_mach_ca"], dtype=np.float64)
    events = {e["name"]: e for e in vehicle["events"]}
    stages = vehicle["stages"]
    # mass dropped at an event: a stage's dry mass at its separation, a dropMass item at its own event
    # (Trajectory_Optimization.py:84-97)
    jettison = {}
    for stage in stages.values():
        if stage.get("separation_at") in events:
            jettison[stage["separation_at"]] = stage["mass_dry"]
        for item in (stage.get("dropMass") or {}).values():
            if item["separation_at"] in events:
                jettison[item["separation_at"]] = item["mass"]
    params = []
    for i, name in enumerate(rows):
        ev = events[name]
        st = stages[str(ev["rocketStage"])]
        params.append({
            "name": name, "time": float(knots[i]),
            "timeFinishAt": float(knots[i + 1]) if i + 1 < len(knots) else float(knots[i]) + 9000.0,
            "rocketStage": ev["rocketStage"], "engineOn": bool(ev["engineOn"]), "thrust": float(ev["thrust"]),
            "nozzle_area": float(ev["nozzle_area"]), "attitude": ev["attitude"],
            "reference_area": float(st["reference_area"]),
            # Trajectory_Optimization.py:109-112
            "massflow": float(ev["thrust"]) / st["Isp_vac"] / 9.80665 if ev["engineOn"] else 0.0,
            "time_ref": ev.get("time_ref"), "mass_jettison": float(jettison.get(name, 0.0)),
        })
    S = len(rows) - 1
    nodes = [int(n) for n in nodes]
    assert len(nodes) == S
    N = sum(nodes)
    pdict = {
        "params": params,
        "ps_params": ps_params if ps_params is not None else PSparams(nodes),
        "wind_table": wind_table, "ca_table": ca_table,
        "N": N, "M": N + S, "num_sections": S, "dx": 1.0e-8,
        # what the knot / time / user constraints read (Trajectory_Optimization.py:116-124)
        "event_index": {p["name"]: i for i, p in enumerate(params)},
        "RocketStage": {k: st for k, st in stages.items()},
    }
    m_init = sum(s["mass_dry"] + s["mass_propellant"] for s in stages.values())
    if vehicle["OptimizationMode"] != "Payload":
        m_init += vehicle["mass_payload"]
    unitdict = {"mass": m_init, "position": 6378137, "velocity": 1000.0, "u": 1.0, "t": params[-1]["time"]}
    # Trajectory_Optimization.py:167-176: terminal targets + the initial state (the launch site in ECI at t = 0, the
    # velocity of the ground there and the launcher's attitude: plain numbers in the vehicle file)
    if "LaunchCondition" in vehicle:
        pdict["LaunchCondition"] = dict(vehicle["LaunchCondition"])      # Trajectory_Optimization.py:103 (downrange origin)
    condition = {"OptimizationMode": vehicle["OptimizationMode"]}
    condition.update(vehicle.get("TerminalCondition", {}))
    condition.update(vehicle.get("FlightConstraint", {}))     # :169 (aero limits, waypoints, antennas)
    if "init" in vehicle:
        condition["init"] = {"mass": m_init, "position": np.array(vehicle["init"]["position"]),
                             "velocity": np.array(vehicle["init"]["velocity"]),
                             "quaternion": np.array(vehicle["init"]["quaternion"]), "u": np.zeros(2)}
        condition["flight_azimuth_init"] = vehicle["LaunchCondition"]["flight_azimuth_init"]
    return pdict, unitdict, condition


_TRAJ_COLS = (["mass"] + ["pos_ECI_" + a for a in "XYZ"] + ["vel_ECI_" + a for a in "XYZ"] +
              ["quat_ECI2BODY_%d" % k for k in range(4)] + ["rate_BODY_Y", "rate_BODY_Z"])


def initial_xdict(vehicle, pdict, unitdict):
    """initialize.py:322-409 (LGR mode): the reference trajectory interpolated at the node times, by the engine's host
    entry point gel_initial_guess (a host-only handle is enough: no GPU is touched)."""
    from .con_dynamics import problem_arrays
    from .engine import Engine
    tab = np.asarray(vehicle["trajectory"], dtype=np.float64)
    col = {c: i for i, c in enumerate(vehicle["trajectory_columns"])}
    ps = pdict["ps_params"]
    S = pdict["num_sections"]
    E = Engine(problem_arrays(pdict, unitdict), D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)], device=-1)
    knots = [p["time"] for p in pdict["params"]]
    x = E.initial_guess(tab[:, col["time"]], tab[:, [col[c] for c in _TRAJ_COLS]], knots)
    out = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in E.split_x(x).items()}
    E.close()
    return out


def synthetic_batch(x0, M, B, seed=20260313):
    """element 0 = x0; element b > 0 = x0 * (1 + 1e-6*noise), position outward only."""
    x0 = np.asarray(x0, dtype=np.float64)
    X = np.tile(x0, (B, 1))
    for b in range(1, B):
        rng = np.random.default_rng(seed + b)
        z = rng.standard_normal(x0.size)
        z[M:4 * M] = np.abs(z[M:4 * M])
        X[b] *= 1.0 + 1e-6 * z
    return X


def make_problem(name="mixed-6x64", vehicle=None, ps_params=None):
    """-> (pdict, unitdict, condition, xdict) for a named configuration."""
    vehicle = vehicle or load_vehicle()
    if name == "example":
        rows, knots, nodes = example_config(vehicle)
    elif name.startswith("stress-12x"):
        rows, knots, nodes = stress_config(vehicle, int(name.split("x")[1]))
    else:
        rows, knots, nodes = CONFIGS[name]
    pdict, unitdict, condition = build_pdict(vehicle, rows, knots, nodes, ps_params)
    if name in CONFIGS:
        # the cut-down event lists of the synthetic meshes do not hold every stage's ignition / cut-off event:
        # inequality_mass leaves those stages out (with a warning) instead of raising like the reference does for a real
        # event list (con_init_terminal_knot.py; lib/con_trajectory.py:40-49)
        pdict["gelato_amd_allow_missing_stage_events"] = True
    return pdict, unitdict, condition, initial_xdict(vehicle, pdict, unitdict)
