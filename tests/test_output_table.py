"""SURVEY.md 8f row f-4, post-processing table (output_result.py:37-263): the numpy oracle against the golden fixture
written from the imported reference, the host share of gelato_amd.output_result (CPU), and the device columns against
the oracle and the fixture (GPU)."""
import numpy as np
import pytest

from conftest import load_golden
from gelato_amd import Engine, problem
from oracle import output_table as ot

XKEYS = ["mass", "position", "velocity", "quaternion", "u", "t"]


def example(device=-1):
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    if device is not None:
        pdict["device"] = device
    return pdict, unitdict, condition, xdict


def xdict_of(x, pdict):
    M, N, S = pdict["M"], pdict["N"], pdict["num_sections"]
    o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N, S + 1])
    return {k: x[o[i]:o[i + 1]].copy() for i, k in enumerate(XKEYS)}


def oracle_table(g, xname, pdict, unitdict):
    S, ps = pdict["num_sections"], pdict["ps_params"]
    params = [(p["thrust"], p["reference_area"], p["nozzle_area"]) for p in pdict["params"]]
    lc = pdict["LaunchCondition"]
    return ot.table(g["x_" + xname], pdict["M"], pdict["N"], [ps.nodes(i) for i in range(S)],
                    (unitdict["mass"], unitdict["position"], unitdict["velocity"]), g["tx_" + xname], params,
                    pdict["wind_table"], pdict["ca_table"], lc["lat"], lc["lon"])


# per-column absolute + relative allowance of the DEVICE values against the reference (device libm against glibc's; the
# altitudes cancel 6.4e6 m, the anomalies / arguments go through acos near its ends)
TOL = {c: (1e-9, 1e-11) for c in Engine.OUTPUT_COLUMNS}
TOL.update({"altitude": (2e-8, 0.0), "altitude_apogee": (1e-5, 1e-12), "altitude_perigee": (1e-5, 1e-12), "downrange": (1e-5, 1e-10),
            "true_anomaly": (2e-6, 0.0), "argument_perigee": (2e-6, 0.0), "lon_ascending_node": (1e-9, 0.0),
            "AOA_total": (2e-6, 1e-9), "AOA_pitch": (1e-9, 1e-9), "AOA_yaw": (1e-9, 1e-9), "Q_alpha": (1e-3, 1e-9),
            "thrust": (1e-6, 1e-12), "aero_BODY_X": (1e-7, 1e-10), "accel_BODY_X": (1e-10, 1e-10),
            "dynamic_pressure": (1e-7, 1e-10), "lat_IIP": (1e-9, 0.0), "lon_IIP": (1e-9, 0.0)})


@pytest.mark.parametrize("xname", ["init", "moved"])
def test_oracle_table_vs_reference_golden(xname):
    g = load_golden("g14_output_table.npz")
    pdict, unitdict, _, _ = example()
    assert list(g["columns"]) == __import__("gelato_amd.output_result", fromlist=["COLUMNS"]).COLUMNS
    T = oracle_table(g, xname, pdict, unitdict)
    n_nan = 0
    for c in ot.DEVICE_COLUMNS:
        ref = g[xname + "_" + c]
        assert np.array_equal(np.isnan(ref), np.isnan(T[c])), c
        n_nan += int(np.isnan(ref).sum())
        # the same formulas on the same libm; the atmosphere is the C oracle's (an ulp from the Python twin's)
        ok = np.abs(T[c] - ref) <= 1e-13 * np.maximum(1.0, np.abs(ref))
        assert np.all(ok | np.isnan(ref)), (c, np.nanmax(np.abs(T[c] - ref)))
    assert n_nan > 0                                               # the orbital end of the trajectory has no impact point
    assert ot.DEVICE_COLUMNS == Engine.OUTPUT_COLUMNS


def test_host_columns_and_node_times_vs_reference_golden():
    """What gelato_amd.output_result forms on the host: node times, text columns, the section of every node, the copies of
    xdict and the interpolated body rates -- bit for bit the reference's (the device share is stubbed out)."""
    from gelato_amd import con_dynamics, output_result as orr
    g = load_golden("g14_output_table.npz")
    pdict, unitdict, _, _ = example()
    E = con_dynamics.engine_of(pdict, unitdict)
    E.output_table = lambda x, tx, la, lo: np.zeros((pdict["M"], len(Engine.OUTPUT_COLUMNS)))   # no GPU here
    for xname in ("init", "moved"):
        xd = xdict_of(g["x_" + xname], pdict)
        tx, tu = orr.node_times(xd, unitdict, pdict)
        # the engine's own LGR nodes are within 1e-14 of the reference's (G1): the times agree to rounding
        assert np.all(np.abs(tx - g["tx_" + xname]) <= 1e-12 * (1.0 + g["tx_" + xname]))
        assert np.all(np.abs(tu - g["tu_" + xname]) <= 1e-12 * (1.0 + g["tu_" + xname]))
        tx, tu = g["tx_" + xname], g["tu_" + xname]
        cols = orr.output_columns(xd, unitdict, tx, tu, pdict)
        assert list(cols) == list(g["columns"])
        for c in cols:
            if c in Engine.OUTPUT_COLUMNS:
                continue
            ref = g[xname + "_" + c]
            got = np.asarray(cols[c])
            if ref.dtype.kind == "U":
                assert [str(v) for v in got] == [str(v) for v in ref], c
            else:
                assert np.array_equal(got, ref), c
        df = orr.output_result(xd, unitdict, tx, tu, pdict)
        assert list(df.columns) == list(g["columns"]) and len(df) == pdict["M"]
    with pytest.raises(Exception):
        Engine.output_table(E, g["x_init"][:-1], g["tx_init"], 0.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("xname", ["init", "moved"])
def test_device_table_vs_reference_golden_and_oracle(xname):
    from gelato_amd import output_result as orr
    g = load_golden("g14_output_table.npz")
    pdict, unitdict, _, _ = example(device=None)
    xd = xdict_of(g["x_" + xname], pdict)
    keep = {k: v.copy() for k, v in xd.items()}
    tx, tu = g["tx_" + xname], g["tu_" + xname]                    # the reference's node times (its own LGR nodes)
    df = orr.output_result(xd, unitdict, tx, tu, pdict)
    assert list(df.columns) == list(g["columns"]) and len(df) == pdict["M"]
    T = oracle_table(g, xname, pdict, unitdict)
    for c in df.columns:
        ref = g[xname + "_" + c]
        got = df[c].to_numpy()
        if ref.dtype.kind == "U":
            assert [str(v) for v in got] == [str(v) for v in ref], c
            continue
        if c not in Engine.OUTPUT_COLUMNS:
            assert np.array_equal(got, ref), c
            continue
        assert np.array_equal(np.isnan(got), np.isnan(ref)), c
        a, r = TOL[c]
        for other, name in ((ref, "reference"), (T[c], "oracle")):
            d = np.abs(got - other)
            assert np.all((d <= a + r * np.abs(other)) | np.isnan(other)), (c, name, np.nanmax(d))
    assert all(np.array_equal(xd[k], keep[k]) for k in xd)
