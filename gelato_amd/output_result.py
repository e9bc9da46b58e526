"""The post-processing table on the GPU.

Drop-in for the reference's output_result.py: ``output_result(xdict, unitdict, tx_res, tu_res, pdict)`` returns the
same pandas DataFrame -- the same 53 columns in the same order, one row per state node (output_result.py:37-263).

The 34 derived columns (geodetic position, impact point, downrange, orbital elements, ground / air velocity, attitude
and flight-path angles, angles of attack, dynamic pressure, Mach number, thrust, axial force and acceleration) come from
one launch of one device thread per node (gel_output_table); the columns the reference copies out of xdict, the
interpolated body rates and the text columns are formed here.  lat_IIP / lon_IIP are NaN where the impact-point
algorithm has no solution, as in the reference (posLLH_IIP_FAA(.., fill_na=False)).
"""
import numpy as np

from . import con_dynamics
from .engine import Engine, pack_x

COLUMNS = ["event", "time", "stage", "section", "thrust", "mass", "lat", "lon", "lat_IIP", "lon_IIP", "downrange", "altitude",
           "altitude_apogee", "altitude_perigee", "inclination", "argument_perigee", "lon_ascending_node", "true_anomaly",
           "pos_ECI_X", "pos_ECI_Y", "pos_ECI_Z", "vel_ECI_X", "vel_ECI_Y", "vel_ECI_Z", "vel_ground_NED_X", "vel_ground_NED_Y",
           "vel_ground_NED_Z", "quat_ECI2BODY_0", "quat_ECI2BODY_1", "quat_ECI2BODY_2", "quat_ECI2BODY_3", "accel_BODY_X",
           "aero_BODY_X", "heading_NED2BODY", "pitch_NED2BODY", "roll_NED2BODY", "vel_inertial",
           "flightpath_vel_inertial_geocentric", "azimuth_vel_inertial_geocentric", "thrust_direction_ECI_X",
           "thrust_direction_ECI_Y", "thrust_direction_ECI_Z", "rate_BODY_X", "rate_BODY_Y", "rate_BODY_Z", "vel_ground",
           "vel_air", "AOA_total", "AOA_pitch", "AOA_yaw", "dynamic_pressure", "Q_alpha", "M"]


def node_times(xdict, unitdict, pdict):
    """(tx_res, tu_res): times [s] of the state nodes and of the LGR nodes (Trajectory_Optimization.py:476-491)"""
    tu, tx = [], []
    for i in range(pdict["num_sections"]):
        to, tf = xdict["t"][i], xdict["t"][i + 1]
        tau = np.asarray(pdict["ps_params"].tau(i))
        tau_x = np.hstack((-1.0, tau))
        tu.append((tau * (tf - to) / 2 + (tf + to) / 2) * unitdict["t"])
        tx.append((tau_x * (tf - to) / 2 + (tf + to) / 2) * unitdict["t"])
    return np.hstack(tx), np.hstack(tu)


def output_columns(xdict, unitdict, tx_res, tu_res, pdict):
    """{column: array} in the reference's order (what its DataFrame is built from)"""
    eng = con_dynamics.engine_of(pdict, unitdict)
    tx_res = np.asarray(tx_res, dtype=np.float64)
    N = len(tx_res)
    dev = eng.output_table(pack_x(xdict), tx_res, pdict["LaunchCondition"]["lat"], pdict["LaunchCondition"]["lon"])
    D = {c: dev[:, k] for k, c in enumerate(Engine.OUTPUT_COLUMNS)}
    mass_ = xdict["mass"] * unitdict["mass"]
    pos_ = xdict["position"].reshape(-1, 3) * unitdict["position"]
    vel_ = xdict["velocity"].reshape(-1, 3) * unitdict["velocity"]
    quat_ = xdict["quaternion"].reshape(-1, 4)
    u_ = xdict["u"].reshape(-1, 2) * unitdict["u"]
    # text columns and the section of every node (:121-143): section s owns its n + 1 state nodes; the last of them
    # carries the name of the event that ends the section
    ps, P = pdict["ps_params"], pdict["params"]
    event, stage = [""] * N, [""] * N
    section = np.zeros(N, dtype="i4")
    event[0] = P[0]["name"]
    i = 0
    for s in range(pdict["num_sections"]):
        n = ps.nodes(s)
        section[i:i + n + 1] = s
        for k in range(i, i + n + 1):
            stage[k] = P[s]["rocketStage"]
        event[i + n] = P[s + 1]["name"]
        i += n + 1
    out = {"event": event, "time": tx_res.round(6), "stage": stage, "section": section, "mass": mass_,
           "pos_ECI_X": pos_[:, 0], "pos_ECI_Y": pos_[:, 1], "pos_ECI_Z": pos_[:, 2],
           "vel_ECI_X": vel_[:, 0], "vel_ECI_Y": vel_[:, 1], "vel_ECI_Z": vel_[:, 2],
           "quat_ECI2BODY_0": quat_[:, 0], "quat_ECI2BODY_1": quat_[:, 1], "quat_ECI2BODY_2": quat_[:, 2],
           "quat_ECI2BODY_3": quat_[:, 3], "vel_inertial": np.linalg.norm(vel_, axis=1), "rate_BODY_X": np.zeros(N),
           "rate_BODY_Y": np.interp(tx_res, tu_res, u_[:, 0]), "rate_BODY_Z": np.interp(tx_res, tu_res, u_[:, 1])}
    out.update(D)
    return {c: out[c] for c in COLUMNS}


def output_result(xdict, unitdict, tx_res, tu_res, pdict):
    """Returns DataFrame that contains optimization results."""
    import pandas as pd
    return pd.DataFrame(output_columns(xdict, unitdict, tx_res, tu_res, pdict))
