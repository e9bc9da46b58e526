"""CPU ORACLE (test infrastructure, NOT the product) for the post-processing table of SURVEY.md 8f row f-4:
output_result.py:37-263 -- per state node: geodetic position, impact point, downrange, orbital elements, ground / air
velocity, attitude angles, angles of attack, dynamic pressure, Mach number, thrust and axial acceleration.

A numpy / math restatement of the reference algorithm, one node per call; every function cites the reference lines it
follows.  Pinned against the reference itself: tests/golden/g14_output_table.npz was written by
tests/golden/make_golden.py from the imported reference module (output_result.py over lib.coordinate, lib.utils,
lib.USStandardAtmosphere, lib.IIP, lib/downrange.py); tests/test_output_table.py checks this file against it.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

The atmosphere comes from the C oracle (oracle/gelato_oracle.c, pinned against the same twins by G2).
"""
import math

import numpy as np

import oracle as _c
from . import waypoint as wp

OMEGA = wp.OMEGA
MU = 3.986004418e14

# the numeric columns the device fills, in the order of the kernel's output rows (gel_output_table)
DEVICE_COLUMNS = ["thrust", "lat", "lon", "lat_IIP", "lon_IIP", "downrange", "altitude", "altitude_apogee", "altitude_perigee",
                  "inclination", "argument_perigee", "lon_ascending_node", "true_anomaly", "vel_ground_NED_X", "vel_ground_NED_Y",
                  "vel_ground_NED_Z", "accel_BODY_X", "aero_BODY_X", "heading_NED2BODY", "pitch_NED2BODY", "roll_NED2BODY",
                  "flightpath_vel_inertial_geocentric", "azimuth_vel_inertial_geocentric", "thrust_direction_ECI_X",
                  "thrust_direction_ECI_Y", "thrust_direction_ECI_Z", "vel_ground", "vel_air", "AOA_total", "AOA_pitch", "AOA_yaw",
                  "dynamic_pressure", "Q_alpha", "M"]


def quatmult(q, p):
    return wp._quatmult(q, p)


def conj(q):
    return wp._conj(q)


def quatrot(q, v):                                    # lib/coordinate.py:55-68
    return quatmult(conj(q), quatmult(np.array((0.0, v[0], v[1], v[2])), q))[1:4]


def ecef2eci(p, t):                                   # :194-214
    c, s = math.cos(OMEGA * t), math.sin(OMEGA * t)
    return np.array([p[0] * c - p[1] * s, p[0] * s + p[1] * c, p[2]])


def quat_eci2ecef(t):                                 # :281-294
    return np.array([math.cos(OMEGA * t / 2.0), 0.0, 0.0, math.sin(OMEGA * t / 2.0)])


def quat_ecef2nedg(pe):                               # :335-359
    la, lo, _ = wp.ecef2geodetic(pe[0], pe[1], pe[2])
    p, l = math.radians(la), math.radians(lo)
    c_hl, s_hl, c_hp, s_hp = math.cos(l / 2.0), math.sin(l / 2.0), math.cos(p / 2.0), math.sin(p / 2.0)
    return np.array([c_hl * (c_hp - s_hp) / math.sqrt(2.0), s_hl * (c_hp + s_hp) / math.sqrt(2.0),
                     -c_hl * (c_hp + s_hp) / math.sqrt(2.0), s_hl * (c_hp - s_hp) / math.sqrt(2.0)])


def quat_eci2nedg(pos, t):                            # :386-397
    return quatmult(quat_eci2ecef(t), quat_ecef2nedg(wp.eci2ecef(pos, t)))


def normalize(v):
    return v / np.linalg.norm(v)


def orbital_elements(r, v):                           # :591-649
    nr = normalize(r)
    c = np.cross(r, v)
    f = np.cross(v, c) - MU * nr
    c1, f1 = normalize(c), normalize(f)
    inc = math.acos(c1[2])
    if inc > 1e-10:
        asc = math.atan2(c1[0], -c1[1])
        n = np.array([math.cos(asc), math.sin(asc), 0.0])
        argp = math.acos(n[0] * f1[0] + n[1] * f1[1])
        if f[2] < 0:
            argp *= -1.0
    else:
        asc = 0.0
        argp = math.atan2(f[1], f[0])
    p = np.linalg.norm(c) ** 2 / MU
    e = np.linalg.norm(f) / MU
    a = p / (1.0 - e ** 2)
    ta = math.acos(f1[0] * nr[0] + f1[1] * nr[1] + f1[2] * nr[2])
    if v[0] * r[0] + v[1] * r[1] + v[2] * r[2] < 0.0:
        ta = 2.0 * np.pi - ta
    if asc < 0.0:
        asc += 2.0 * np.pi
    if argp < 0.0:
        argp += 2.0 * np.pi
    if ta < 0.0:
        ta += 2.0 * np.pi
    return np.array([a, e, math.degrees(inc), math.degrees(asc), math.degrees(argp), math.degrees(ta)])


distance_vincenty = wp.distance_vincenty        # lib/downrange.py:32-111 (restated in oracle/waypoint.py: the downrange rows use it too)


def euler_from_quat(q):                               # lib/coordinate.py:505-528
    if 2.0 * (q[0] * q[2] - q[3] * q[1]) >= 1.0:
        el, az, ro = np.pi / 2, 0.0, 0.0
    else:
        az = math.atan2(2.0 * (q[0] * q[3] + q[1] * q[2]), 1.0 - 2.0 * (q[2] ** 2 + q[3] ** 2))
        el = math.asin(2.0 * (q[0] * q[2] - q[3] * q[1]))
        ro = math.atan2(2.0 * (q[0] * q[1] + q[2] * q[3]), 1.0 - 2.0 * (q[1] ** 2 + q[2] ** 2))
    if az < 0.0:
        az += 2.0 * np.pi
    return np.rad2deg(np.array([az, el, ro]))


def posLLH_IIP_nan(pe, ve):
    """posLLH_IIP_FAA(.., fill_na=False) (lib/IIP.py:38-41): NaN where the algorithm has no solution.  The filled form
    returns exact zeros there and nowhere else on a real trajectory."""
    r = wp.posLLH_IIP_FAA(pe, ve)
    return np.full(3, np.nan) if (r[0] == 0.0 and r[1] == 0.0) else r


def _air_velocity_eci(pos, vel, t, wind_table):
    """lib/utils.py:105-114 (both angle-of-attack functions; altitude from the ECI position, as they do)"""
    llh = wp.ecef2geodetic(pos[0], pos[1], pos[2])
    h = _c.geopotential_altitude(llh[2])
    w = np.array([*_c.wind_ned(h, wind_table)[:2], 0.0])
    vel_ecef = wp.vel_eci2ecef(vel, pos, t)
    w_eci = quatrot(conj(quat_eci2nedg(pos, t)), w)
    return ecef2eci(vel_ecef, t) - w_eci


def aoa_all_rad(pos, vel, quat, t, wind_table):       # lib/utils.py:92-121
    tdir = quatrot(conj(quat), np.array([1.0, 0.0, 0.0]))
    va = _air_velocity_eci(pos, vel, t, wind_table)
    c = normalize(va).dot(normalize(tdir))
    return 0.0 if (c >= 1.0 or np.linalg.norm(va) < 0.001) else math.acos(c)


def aoa_ab_rad(pos, vel, quat, t, wind_table):        # :132-161
    vb = quatrot(quat, _air_velocity_eci(pos, vel, t, wind_table))
    if vb[0] < 0.001:
        return np.zeros(2)
    return np.array((math.atan2(vb[2], vb[0]), math.atan2(vb[1], vb[0])))


def node_row(mass, pos, vel, quat_raw, t, par, wind_table, ca_table, launch_lat, launch_lon):
    """output_result.py:126-262 for one node: {column: value} of DEVICE_COLUMNS.  par = (thrust, reference_area,
    nozzle_area) of the node's section; units already applied (kg, m, m/s, s)."""
    o = {}
    quat = normalize(quat_raw)
    thrust_vac, area, nozzle = par
    llh = wp.eci2geodetic(pos, t)
    h = _c.geopotential_altitude(llh[2])
    o["lat"], o["lon"], o["altitude"] = llh
    o["downrange"] = distance_vincenty(launch_lat, launch_lon, llh[0], llh[1])
    el = orbital_elements(pos, vel)
    o["altitude_apogee"] = el[0] * (1.0 + el[1]) - 6378137
    o["altitude_perigee"] = el[0] * (1.0 - el[1]) - 6378137
    o["inclination"], o["lon_ascending_node"], o["argument_perigee"], o["true_anomaly"] = el[2:6]
    vg_ecef = wp.vel_eci2ecef(vel, pos, t)
    vg_ned = quatrot(quat_ecef2nedg(wp.eci2ecef(pos, t)), vg_ecef)
    o["vel_ground_NED_X"], o["vel_ground_NED_Y"], o["vel_ground_NED_Z"] = vg_ned
    v_ned = quatrot(quat_eci2nedg(pos, t), vel)
    w_ned = np.array([*_c.wind_ned(h, wind_table)[:2], 0.0])
    va_ned = vg_ned - w_ned
    o["vel_ground"] = np.linalg.norm(vg_ecef)
    o["azimuth_vel_inertial_geocentric"] = math.degrees(math.atan2(v_ned[1], v_ned[0]))
    o["flightpath_vel_inertial_geocentric"] = math.degrees(math.asin(-v_ned[2] / np.linalg.norm(v_ned)))
    rho = _c.air_density(h)
    q = 0.5 * np.linalg.norm(va_ned) ** 2 * rho
    o["dynamic_pressure"] = q
    a_all = aoa_all_rad(pos, vel, quat, t, wind_table) * 180.0 / np.pi
    a_ab = aoa_ab_rad(pos, vel, quat, t, wind_table) * 180.0 / np.pi
    o["AOA_total"], o["Q_alpha"] = a_all, a_all * q
    o["AOA_pitch"], o["AOA_yaw"] = a_ab
    tdir = quatrot(conj(quat), np.array([1.0, 0.0, 0.0]))
    o["thrust_direction_ECI_X"], o["thrust_direction_ECI_Y"], o["thrust_direction_ECI_Z"] = tdir
    eu = euler_from_quat(quatmult(conj(quat_eci2nedg(pos, t)), quat))       # quat_nedg2body (:488-502)
    o["heading_NED2BODY"], o["pitch_NED2BODY"], o["roll_NED2BODY"] = eu
    p = _c.air_pressure(h)
    pos_ecef = wp.eci2ecef(pos, t)
    vel_ecef = wp.vel_eci2ecef(vel, pos, t)
    w_eci = quatrot(conj(quat_eci2nedg(pos, t)), w_ned)
    va_eci = ecef2eci(vel_ecef, t) - w_eci
    mach = np.linalg.norm(va_eci) / _c.speed_of_sound(h)
    o["M"] = mach
    ca = np.interp(mach, ca_table[:, 0], ca_table[:, 1])
    o["vel_air"] = np.linalg.norm(va_eci)
    aero_eci = 0.5 * rho * np.linalg.norm(va_eci) * -va_eci * area * ca
    aero_body = quatrot(quat, aero_eci)
    thrust = thrust_vac - nozzle * p
    o["thrust"] = thrust
    o["aero_BODY_X"] = aero_body[0]
    o["accel_BODY_X"] = (thrust + aero_body[0]) / mass
    o["lat_IIP"], o["lon_IIP"], _ = posLLH_IIP_nan(pos_ecef, vel_ecef)
    return o


def node_sections(nodes):
    """section of every state node (output_result.py:121-143): the n + 1 state nodes of section s"""
    return np.concatenate([np.full(n + 1, s, dtype=np.int32) for s, n in enumerate(nodes)])


def table(x, M, N, nodes, units, tx_res, params, wind_table, ca_table, launch_lat, launch_lon):
    """{column: [M]} of DEVICE_COLUMNS; params[s] = (thrust, reference_area, nozzle_area); units = (mass, pos, vel)"""
    from .knot_terminal import split
    mass, pos, vel, quat, _, _ = split(np.asarray(x, dtype=np.float64), M, N)
    sec = node_sections(nodes)
    rows = [node_row(mass[i] * units[0], pos[i] * units[1], vel[i] * units[2], quat[i], tx_res[i], params[sec[i]], wind_table,
                     ca_table, launch_lat, launch_lon) for i in range(M)]
    return {c: np.array([r[c] for r in rows]) for c in DEVICE_COLUMNS}
