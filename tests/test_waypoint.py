"""SURVEY.md 8f row f-4, waypoint rows (lib/con_waypoint.py): the numpy oracle against the golden fixture written from
the imported reference, the host logic of gelato_amd.con_waypoint against both (CPU), and the device rows against the
oracle and the fixture (GPU)."""
import json

import numpy as np
import pytest

from conftest import load_golden
from gelato_amd import Engine, problem
from oracle import knot_terminal as kt
from oracle import waypoint as wp

GROUPS = ("eqpos", "ineqpos", "eqiip", "ineqiip", "antenna")
XKEYS = ["mass", "position", "velocity", "quaternion", "u", "t"]


def example(extra=None, device=-1):
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    if device is not None:
        pdict["device"] = device                          # -1: host-only handle, describes but never evaluates
    return pdict, unitdict, dict(condition, **(extra or {})), xdict


def conditions(g):
    return json.loads(str(g["conds_json"]))


def xdict_of(x, pdict):
    M, N, S = pdict["M"], pdict["N"], pdict["num_sections"]
    o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N, S + 1])
    return {k: x[o[i]:o[i + 1]].copy() for i, k in enumerate(XKEYS)}


# ------------------------------------------------------------------ oracle vs the reference's own outputs
@pytest.mark.parametrize("cname", ["example", "synthetic"])
@pytest.mark.parametrize("xname", ["init", "moved"])
def test_oracle_rows_vs_reference_golden(xname, cname):
    g = load_golden("g13_waypoint.npz")
    pdict, unitdict, condition, _ = example(conditions(g)[cname])
    sp = kt.make_spec(pdict, unitdict, condition)
    rows = wp.make_rows(sp, pdict, condition)
    x = g["x_" + xname]
    seen = 0
    for grp in GROUPS:
        base = "%s_%s_%s" % (xname, cname, grp)
        v = wp.values(x, sp, rows, grp)
        if bool(g[base + "_none"]):
            assert v is None and wp.jacobian(x, sp, rows, grp) is None
            continue
        seen += 1
        assert np.array_equal(v, g[base + "_con"]), base           # the same libm, the same order of operations: bits
        J, xm = wp.jacobian(x, sp, rows, grp, drift=True)          # with the reference's in-place perturbation
        J0 = wp.jacobian(x, sp, rows, grp)                         # with the product's semantics (x untouched)
        assert np.max(np.abs(xm - x)) <= 2.3e-16
        for var, (r, c, vals, shape) in J.items():
            k = base + "_jac_" + var
            assert np.array_equal(r, g[k + "_rows"]) and np.array_equal(c, g[k + "_cols"]) and r.dtype == np.int32, k
            assert shape == tuple(g[k + "_shape"]) and np.array_equal(vals, g[k + "_vals"]), k
            assert np.all(np.abs(J0[var][2] - vals) <= 1e-5 + 1e-6 * np.abs(vals)), k
        assert sorted(J) == sorted(k[len(base) + 5:-5] for k in g if k.startswith(base + "_jac_") and k.endswith("_rows"))
    assert seen == (3 if cname == "example" else 5)


def test_oracle_point_functions_vs_reference_golden():
    g = load_golden("g13_waypoint.npz")
    pos, vel, tt = g["pt_pos"], g["pt_vel"], g["pt_t"]
    assert np.array_equal(np.array([wp.eci2geodetic(p, t) for p, t in zip(pos, tt)]), g["pt_geodetic"])
    iip = np.array([wp.posLLH_IIP_FAA(wp.eci2ecef(p, t), wp.vel_eci2ecef(v, p, t)) for p, v, t in zip(pos, vel, tt)])
    assert np.array_equal(iip, g["pt_iip"])
    none = np.all(iip == 0.0, axis=1)                              # the orbital end of the trajectory has no impact point
    assert 0 < none.sum() < len(none) and none[-1] and not none[0]
    assert np.array_equal(wp.geodetic2ecef(42.50587, 143.45659, 50.0), g["pt_ant_ecef"])
    pdict, unitdict, condition, _ = example()
    sp = kt.make_spec(pdict, unitdict, condition)
    se = [wp._f_elev(p / sp["units"]["position"], None, t / sp["units"]["t"], sp, g["pt_ant_ecef"])[0] for p, t in zip(pos, tt)]
    assert np.array_equal(np.array(se), g["pt_sin_elev"])


# ------------------------------------------------------------------ host logic of the product, no GPU
def mode_value(f, row):
    """The arithmetic the kernel applies to the function value (include/gelato_amd.h, node-function rows)."""
    _, _, _, mode, p = row
    v = (f - p[1]) / p[0] if mode & 1 else f / p[0] - p[1]
    return -v if mode & 8 else v


@pytest.mark.parametrize("cname", ["example", "synthetic"])
def test_product_row_table_reproduces_the_reference_values(cname):
    """gelato_amd.con_waypoint.build_rows: groups, nodes, knot-time columns, functions, modes and parameters such that the
    reference's values come out bit for bit when the function values are the oracle's."""
    from gelato_amd import con_init_terminal_knot as ck
    g = load_golden("g13_waypoint.npz")
    pdict, unitdict, condition, _ = example(conditions(g)[cname])
    R = ck.rows_of(pdict, unitdict, condition)
    sp = kt.make_spec(pdict, unitdict, condition)
    orows = wp.make_rows(sp, pdict, condition)
    assert [r[0] for r in R.waypoint_rows] == [r[0] for r in sorted(orows, key=lambda r: GROUPS.index(r[0]))]
    assert R.fn[R.waypoint_base:] == [r[3] for r in R.waypoint_rows] and R.nfn == len(R.fn)
    fam = {"latitude_deg": (wp._f_llh, 0), "longitude_deg": (wp._f_llh, 1), "altitude": (wp._f_llh, 2),
           "lat_IIP_deg": (wp._f_iip, 0), "lon_IIP_deg": (wp._f_iip, 1), "sin_elevation": (wp._f_elev, 0)}
    for xname in ("init", "moved"):
        x = g["x_" + xname]
        _, pos, vel, _, _, t = kt.split(x, sp["M"], sp["N"])
        for grp in GROUPS:
            a, b = R.waypoint_slices[grp]
            base = "%s_%s_%s" % (xname, cname, grp)
            if bool(g[base + "_none"]):
                assert a == b
                continue
            vals = []
            for (_, sec, node, row) in R.waypoint_rows[a:b]:
                assert row[1] == node == sp["xa"][sec] and row[2] == sec and row[3] & 4
                f, comp = fam[row[0]]
                ant = np.array(row[4][2:5]) if row[0] == "sin_elevation" else None
                if ant is not None:                                # the antenna's vertical: the ellipsoid normal
                    assert np.allclose(row[4][5:8], wp.antenna_vertical(ant), rtol=0, atol=1e-15)
                vals.append(mode_value(f(pos[node], vel[node], t[sec], sp, ant)[comp], row))
            ref = g[base + "_con"]
            if grp == "antenna":                                   # the vertical is formed differently: an ulp of sin(elevation)
                assert np.all(np.abs(np.array(vals) - ref) <= 4e-16), base
            else:
                assert np.array_equal(np.array(vals), ref), base
    # the C-ABI accepted the table (host-only handle: validation, no device work)
    E = R.engine
    assert E._nfn == R.nfn and E._nlin == R.nlin


def test_product_returns_none_without_rows_and_needs_a_min_bound_for_downrange_max():
    from gelato_amd import con_waypoint as cw
    # the reference's downrange `max` row divides by the `min` bound (con_waypoint.py:778): without one it raises KeyError
    pdict, unitdict, condition, xdict = example({"waypoint": {"FAIRING": {"downrange": {"max": 1.0e5}}}, "antenna": {}})
    with pytest.raises(KeyError, match="min"):
        cw.inequality_posLLH(xdict, pdict, unitdict, condition)
    pdict, unitdict, condition, xdict = example()
    condition = {k: v for k, v in condition.items() if k not in ("waypoint", "antenna")}
    for f in (cw.equality_posLLH, cw.equality_jac_posLLH, cw.inequality_posLLH, cw.inequality_jac_posLLH, cw.equality_IIP,
              cw.equality_jac_IIP, cw.inequality_IIP, cw.inequality_jac_IIP, cw.inequality_antenna, cw.inequality_jac_antenna):
        assert f(xdict, pdict, unitdict, condition) is None        # con_waypoint.py:82-83,178-179,...: no table, no rows
    # groups without rows are None while others exist (the shipped example: no equality IIP, no inequality position rows)
    pdict, unitdict, condition, xdict = example()
    from gelato_amd import con_init_terminal_knot as ck
    R = ck.rows_of(pdict, unitdict, condition)
    assert R.waypoint_slices["eqiip"] == (0, 0) and R.waypoint_slices["ineqpos"] == (0, 0)
    assert [r[0] for r in R.waypoint_rows] == ["eqpos", "ineqiip", "antenna"]
    with pytest.raises(Exception, match="host-only"):
        cw.equality_posLLH(xdict, pdict, unitdict, condition)


def test_row_abi_validation():
    pdict, unitdict, condition, _ = example()
    from gelato_amd import con_dynamics
    E = con_dynamics.engine_of(pdict, unitdict)
    ok = ("latitude_deg", 3, 1, Engine.MODE_SHIFTED | Engine.MODE_RAW_DIFFERENCE, [90.0, 42.0])
    E.rows_configure([], [ok])
    for bad in (("latitude_deg", 3, -1, 5, [90.0, 42.0]),          # a function of the knot time without a knot-time column
                ("latitude_deg", 3, 1, 16, [90.0, 42.0]),          # unknown mode bit
                ("latitude_deg", 3, 1, 2, [90.0, 42.0]),           # unknown value form
                ("latitude_deg", 3, pdict["num_sections"] + 1, 5, [90.0, 42.0]),
                (16, 3, 1, 5, [90.0, 42.0]),
                ("altitude", 3, 1, 4, [0.0, 1.0])):                # zero scale
        with pytest.raises(Exception):
            E.rows_configure([], [bad])


# ------------------------------------------------------------------ the "downrange" rows (G13b)
@pytest.mark.parametrize("cname", ["dr", "dronly"])
@pytest.mark.parametrize("xname", ["init", "moved"])
def test_oracle_downrange_rows_vs_reference_golden(xname, cname):
    """Values bit for bit; Jacobian numbers bit for bit against the reference's RAW (scrambled) lists: it appends a downrange
    row's t value to the position values (con_waypoint.py:702-706,915-919,932-936)."""
    g = load_golden("g13b_downrange.npz")
    cd = conditions(g)[cname]
    pdict, unitdict, condition, _ = example(dict(cd, antenna={}))
    sp = kt.make_spec(pdict, unitdict, condition)
    rows = wp.make_rows(sp, pdict, condition)
    names = [pdict["params"][i]["name"] for i in range(pdict["num_sections"])]
    x = g["x_" + xname]
    seen = 0
    for grp in ("eqpos", "ineqpos"):
        base = "%s_%s_%s" % (xname, cname, grp)
        v = wp.values(x, sp, rows, grp)
        if bool(g[base + "_none"]):
            assert v is None
            continue
        seen += 1
        assert np.array_equal(v, g[base + "_con"]), base
        J, _ = wp.jacobian(x, sp, rows, grp, drift=True)
        mine = [r for r in rows if r[0] == grp]
        # index lists: what the reference emits (three position pairs and one t pair per row)
        assert np.array_equal(J["position"][0], g[base + "_jac_position_rows"]) and np.array_equal(J["position"][1], g[base + "_jac_position_cols"])
        assert np.array_equal(J["t"][0], g[base + "_jac_t_rows"]) and np.array_equal(J["t"][1], g[base + "_jac_t_cols"])
        bounds = {ir: (cd["waypoint"][names[r[1]]]["downrange"]["min"], cd["waypoint"][names[r[1]]]["downrange"]["max"])
                  for ir, r in enumerate(mine) if r[3] == "dr" and r[5] == "max"}
        pv, tv = wp.reference_downrange_lists(J, rows, grp, bounds)
        rp, rt = g[base + "_jac_position_vals"], g[base + "_jac_t_vals"]
        ndr = sum(r[3] == "dr" for r in mine)
        assert len(rp) == 3 * len(mine) + ndr and len(rt) == len(mine) - ndr       # the scramble, as documented
        assert np.array_equal(tv, rt), base
        exact = np.ones(len(pv), dtype=bool)
        k = 0
        for ir, r in enumerate(mine):                                              # the re-scaled t value of a max row: one rounding
            k += 3
            if r[3] == "dr":
                exact[k] = r[5] != "max"
                k += 1
        assert np.array_equal(pv[exact], rp[exact]), base
        assert np.all(np.abs(pv[~exact] - rp[~exact]) <= 4e-16 * np.abs(rp[~exact])), base
        J0 = wp.jacobian(x, sp, rows, grp)                                         # the product's semantics: x untouched
        for var in ("position", "t"):
            assert np.all(np.abs(J0[var][2] - J[var][2]) <= 1e-5 + 1e-6 * np.abs(J[var][2])), (base, var)
    assert seen == (2 if cname == "dr" else 1)


def test_oracle_downrange_gradient_vs_reference_golden():
    g = load_golden("g13b_downrange.npz")
    pdict, unitdict, condition, _ = example()
    sp = kt.make_spec(pdict, unitdict, condition)
    origin = tuple(g["launch_lat_lon"])
    dx = sp["dx"]
    for p_, t_, f0, gp, gt in zip(g["pt_pos"], g["pt_t"], g["pt_downrange"], g["pt_grad_position"], g["pt_grad_t"]):
        fc = wp._f_downrange(p_, None, t_, sp, origin)[0]
        assert fc == f0
        q = p_.copy()
        for j in range(3):                                                          # con_waypoint.py:597-601, in place
            q[j] += dx
            assert (wp._f_downrange(q, None, t_, sp, origin)[0] - fc) / dx == gp[j]
            q[j] -= dx
        assert (wp._f_downrange(q, None, t_ + dx, sp, origin)[0] - fc) / dx == gt


@pytest.mark.parametrize("cname", ["dr", "dronly"])
def test_product_row_table_reproduces_the_reference_downrange_values(cname):
    from gelato_amd import con_init_terminal_knot as ck
    g = load_golden("g13b_downrange.npz")
    pdict, unitdict, condition, _ = example(dict(conditions(g)[cname], antenna={}))
    R = ck.rows_of(pdict, unitdict, condition)
    sp = kt.make_spec(pdict, unitdict, condition)
    orows = wp.make_rows(sp, pdict, condition)
    assert [(r[0], r[1]) for r in R.waypoint_rows] == [(r[0], r[1]) for r in sorted(orows, key=lambda r: GROUPS.index(r[0]))]
    fam = {"latitude_deg": (wp._f_llh, 0), "longitude_deg": (wp._f_llh, 1), "altitude": (wp._f_llh, 2), "downrange": (wp._f_downrange, 0)}
    assert sum(r[3][0] == "downrange" for r in R.waypoint_rows) == (6 if cname == "dr" else 1)
    for xname in ("init", "moved"):
        x = g["x_" + xname]
        _, pos, vel, _, _, t = kt.split(x, sp["M"], sp["N"])
        for grp in ("eqpos", "ineqpos"):
            a, b = R.waypoint_slices[grp]
            base = "%s_%s_%s" % (xname, cname, grp)
            if bool(g[base + "_none"]):
                assert a == b
                continue
            vals = []
            for (_, sec, node, row) in R.waypoint_rows[a:b]:
                f, comp = fam[row[0]]
                aux = tuple(row[4][2:4]) if row[0] == "downrange" else None
                if aux is not None:
                    assert aux == tuple(g["launch_lat_lon"]) and row[3] & 4 and row[2] == sec
                vals.append(mode_value(f(pos[node], vel[node], t[sec], sp, aux)[comp], row))
            assert np.array_equal(np.array(vals), g[base + "_con"]), base


# ------------------------------------------------------------------ the device rows
@pytest.mark.gpu
@pytest.mark.parametrize("cname", ["example", "synthetic"])
def test_device_waypoint_functions_vs_reference_golden(cname):
    from gelato_amd import con_waypoint as cw
    g = load_golden("g13_waypoint.npz")
    pdict, unitdict, condition, _ = example(conditions(g)[cname], device=None)
    fns = {"eqpos": (cw.equality_posLLH, cw.equality_jac_posLLH), "ineqpos": (cw.inequality_posLLH, cw.inequality_jac_posLLH),
           "eqiip": (cw.equality_IIP, cw.equality_jac_IIP), "ineqiip": (cw.inequality_IIP, cw.inequality_jac_IIP),
           "antenna": (cw.inequality_antenna, cw.inequality_jac_antenna)}
    for xname in ("init", "moved"):
        xd = xdict_of(g["x_" + xname], pdict)
        keep = {k: v.copy() for k, v in xd.items()}
        for grp, (f, jf) in fns.items():
            base = "%s_%s_%s" % (xname, cname, grp)
            con, jac = f(xd, pdict, unitdict, condition), jf(xd, pdict, unitdict, condition)
            if bool(g[base + "_none"]):
                assert con is None and jac is None
                continue
            ref = g[base + "_con"]
            # device libm (sincos, atan2, asin, tan) against glibc's: a few ulp of angles of O(100) degrees, scaled by 1/90
            assert con.shape == ref.shape and np.all(np.abs(con - ref) <= 1e-12), (base, np.abs(con - ref).max())
            assert sorted(jac) == sorted(k[len(base) + 5:-5] for k in g if k.startswith(base + "_jac_") and k.endswith("_rows"))
            for var, blk in jac.items():
                k = base + "_jac_" + var
                assert np.array_equal(blk["coo"][0], g[k + "_rows"]) and np.array_equal(blk["coo"][1], g[k + "_cols"]), k
                assert blk["coo"][0].dtype == np.int32 and blk["shape"] == tuple(g[k + "_shape"]), k
                rv = g[k + "_vals"]
                # forward differences with dx = 1e-8: an ulp of the function value is worth 1e-8 / dx of it here
                assert np.all(np.abs(blk["coo"][2] - rv) <= 2e-5 + 1e-6 * np.abs(rv)), (k, np.abs(blk["coo"][2] - rv).max())
        assert all(np.array_equal(xd[k], keep[k]) for k in xd)     # xdict is never mutated


@pytest.mark.gpu
def test_device_rows_batch_vs_oracle_including_states_without_an_impact_point():
    """Every function (fn 9-14) at every knot the rows may name, over a batch of moved decision vectors: values and
    seven-column differences against the numpy oracle; knots on the orbital end of the trajectory have no impact point
    (value from (0, 0), zero differences), as lib/IIP.py fills them."""
    pdict, unitdict, condition, xdict = example(device=None)
    from gelato_amd import con_dynamics
    from gelato_amd.engine import pack_x
    E = con_dynamics.engine_of(pdict, unitdict)
    sp = kt.make_spec(pdict, unitdict, condition)
    S = pdict["num_sections"]
    ant = wp.geodetic2ecef(36.0, 140.0, 300.0)
    up = wp.antenna_vertical(ant)
    SH, RAW, NEG = Engine.MODE_SHIFTED, Engine.MODE_RAW_DIFFERENCE, Engine.MODE_NEGATED
    rows, meta = [], []
    for sec in range(S):
        node = sp["xa"][sec]
        for f, fam, comp, mode, p in (("latitude_deg", wp._f_llh, 0, SH | RAW, [90.0, 41.0]), ("longitude_deg", wp._f_llh, 1, SH | RAW | NEG, [180.0, 150.0]),
                                      ("altitude", wp._f_llh, 2, RAW | NEG, [1.0e5, 1.0]), ("lat_IIP_deg", wp._f_iip, 0, SH | RAW | NEG, [90.0, 30.0]),
                                      ("lon_IIP_deg", wp._f_iip, 1, SH | RAW, [180.0, 160.0]),
                                      ("sin_elevation", wp._f_elev, 0, RAW, [1.0, 0.1, *ant, *up])):
            rows.append((f, node, sec, mode, p))
            meta.append((fam, comp, node, sec, mode, p))
    E.rows_configure([], rows)
    rng = np.random.default_rng(5)
    x0 = pack_x(xdict)
    B = 6
    X = x0[None, :] * (1.0 + 1e-3 * rng.standard_normal((B, x0.size)))
    X[:, -(S + 1):] = np.sort(X[:, -(S + 1):], axis=1)
    con, jfn, rc = E.rows_eval(X)
    assert rc == 0 and jfn.shape == (B, len(rows), 7)
    dx = sp["dx"]
    n_none = 0
    for b in range(B):
        _, pos, vel, _, _, t = kt.split(X[b], sp["M"], sp["N"])
        for r, (fam, comp, node, sec, mode, p) in enumerate(meta):
            ecef = np.array(p[2:5]) if fam is wp._f_elev else None
            fc = fam(pos[node], vel[node], t[sec], sp, ecef)[comp]
            v = mode_value(fc, (None, None, None, mode, p))
            tol = 1e-12 if comp != 2 or fam is not wp._f_llh else 1e-11
            assert abs(con[b, r] - v) <= tol, (b, r, con[b, r], v)
            d = np.zeros(7)
            for c in range(7):
                pp, vv, tt = pos[node].copy(), vel[node].copy(), t[sec]
                if c < 3:
                    pp[c] += dx
                elif c < 6:
                    vv[c - 3] += dx
                else:
                    tt = tt + dx
                d[c] = ((fam(pp, vv, tt, sp, ecef)[comp] - fc) / dx) / p[0] * (-1.0 if mode & 8 else 1.0)
            if fam is not wp._f_iip:
                assert np.all(jfn[b, r, 3:6] == 0.0)               # position-only functions: exact zeros for velocity
            if fam is wp._f_iip and fc == 0.0 and fam(pos[node], vel[node], t[sec], sp, None)[1 - comp] == 0.0:
                n_none += 1
                assert np.all(jfn[b, r] == 0.0) and con[b, r] == v  # no impact point: (0, 0), bit for bit
            scale = 2e-5 if (fam is wp._f_llh and comp == 2) else 2e-6
            assert np.all(np.abs(jfn[b, r] - d) <= scale + 1e-6 * np.abs(d)), (b, r, jfn[b, r], d)
    assert n_none > 0
    # one decision vector through the callback's single round trip gives the same bits
    fr = E.eval_callback(X[2], True)
    assert np.array_equal(fr["rows_con"], con[2]) and np.array_equal(fr["rows_jfn"], jfn[2])


@pytest.mark.gpu
@pytest.mark.parametrize("cname", ["dr", "dronly"])
def test_device_downrange_rows_vs_reference_golden(cname):
    """The downrange rows through the reference-named functions: values against the reference's; Jacobian numbers against the
    reference's raw lists, unscrambled (position values of a downrange row = entries 0-2 of its four, t value = the fourth;
    the max row's t value re-scaled from the reference's -1/max to the -1/min of its value and position entries)."""
    from gelato_amd import con_waypoint as cw
    g = load_golden("g13b_downrange.npz")
    cd = conditions(g)[cname]
    pdict, unitdict, condition, _ = example(dict(cd, antenna={}), device=None)
    sp = kt.make_spec(pdict, unitdict, condition)
    rows = wp.make_rows(sp, pdict, condition)
    names = [pdict["params"][i]["name"] for i in range(pdict["num_sections"])]
    dx = sp["dx"]
    for xname in ("init", "moved"):
        xd = xdict_of(g["x_" + xname], pdict)
        for grp, f, jf in (("eqpos", cw.equality_posLLH, cw.equality_jac_posLLH), ("ineqpos", cw.inequality_posLLH, cw.inequality_jac_posLLH)):
            base = "%s_%s_%s" % (xname, cname, grp)
            con, jac = f(xd, pdict, unitdict, condition), jf(xd, pdict, unitdict, condition)
            if bool(g[base + "_none"]):
                assert con is None and jac is None
                continue
            mine = [r for r in rows if r[0] == grp]
            ref = g[base + "_con"]
            # Vincenty's iteration stops at |d lambda| < 1e-12 rad (lib/downrange.py:93): two libms may stop one trip apart,
            # 1e-12 rad of longitude = 6.4e-6 m of distance
            tolv = np.array([6.4e-6 / r[6] + 1e-12 if r[3] == "dr" else 1e-12 for r in mine])
            assert con.shape == ref.shape and np.all(np.abs(con - ref) <= tolv), (base, np.abs(con - ref).max())
            assert sorted(jac) == ["position", "t"]
            assert np.array_equal(jac["position"]["coo"][0], g[base + "_jac_position_rows"]) and np.array_equal(jac["position"]["coo"][1], g[base + "_jac_position_cols"])
            assert np.array_equal(jac["t"]["coo"][0], g[base + "_jac_t_rows"]) and np.array_equal(jac["t"]["coo"][1], g[base + "_jac_t_cols"])
            rp, rt = list(g[base + "_jac_position_vals"]), list(g[base + "_jac_t_vals"])
            for ir, r in enumerate(mine):
                pos3 = [rp.pop(0) for _ in range(3)]
                if r[3] == "dr":
                    tval = rp.pop(0)
                    if r[5] == "max":
                        b = cd["waypoint"][names[r[1]]]["downrange"]
                        tval = tval * b["max"] / b["min"]
                    # forward difference of a distance of up to 2e6 m: an ulp of it is worth 4e-10 / dx / bound; the stop
                    # criterion of the iteration adds up to 6.4e-6 m / dx / bound when the two evaluations stop a trip apart
                    tol = (4e-10 + 6.4e-6) / dx / r[6]
                else:
                    tval = rt.pop(0)
                    tol = 2e-5
                got = list(jac["position"]["coo"][2][3 * ir:3 * ir + 3]) + [jac["t"]["coo"][2][ir]]
                for a_, b_ in zip(got, pos3 + [tval]):
                    assert abs(a_ - b_) <= tol + 1e-6 * abs(b_), (base, ir, a_, b_, tol)
            assert not rp and not rt


@pytest.mark.gpu
def test_one_row_groups_return_fresh_jacobian_arrays():
    """A group with exactly ONE row: `jfn[a:a+1, 0:3]` is contiguous, so ravel() is a view of the engine's buffer, which the next
    callback overwrites.  The values returned for the first decision vector must still be there after a second one
    has been evaluated."""
    from gelato_amd import con_waypoint as cw
    cond = {"waypoint": {"SEIG": {"altitude": {"exact": 1.0e5}, "lat_IIP": {"min": 20.0}}},
            "antenna": {"ANT": {"lon": 143.45659, "lat": 42.50587, "altitude": 50.0, "elevation_min": {"SECO": 0.0}}}}
    pdict, unitdict, condition, xdict = example(cond, device=None)
    x2 = {k: v * (1.0 + 1e-3) if k in ("position", "velocity") else v.copy() for k, v in xdict.items()}
    for jf in (cw.equality_jac_posLLH, cw.inequality_jac_IIP, cw.inequality_jac_antenna):
        first = jf(xdict, pdict, unitdict, condition)
        keep = {var: blk["coo"][2].copy() for var, blk in first.items()}
        assert all(len(v) in (1, 3) for v in keep.values())            # one row: three position / velocity values, one t value
        second = jf(x2, pdict, unitdict, condition)
        for var in keep:
            assert np.array_equal(first[var]["coo"][2], keep[var]), (jf.__name__, var, "the first call's values changed under the second")
            assert not np.shares_memory(first[var]["coo"][2], second[var]["coo"][2])
        assert any(not np.array_equal(second[var]["coo"][2], keep[var]) for var in keep)
    # shared value arrays on request (like the defect groups'): the same numbers, in arrays that the next call rewrites in place
    pdict["gelato_amd_share_values"] = True
    for jf in (cw.equality_jac_posLLH, cw.inequality_jac_IIP, cw.inequality_jac_antenna):
        pdict["gelato_amd_share_values"] = False
        fresh1, fresh2 = jf(xdict, pdict, unitdict, condition), jf(x2, pdict, unitdict, condition)
        pdict["gelato_amd_share_values"] = True
        s1 = jf(xdict, pdict, unitdict, condition)
        assert list(s1) == list(fresh1) and all(np.array_equal(s1[v]["coo"][2], fresh1[v]["coo"][2]) and s1[v]["shape"] == fresh1[v]["shape"]
                                                and np.array_equal(s1[v]["coo"][0], fresh1[v]["coo"][0]) and np.array_equal(s1[v]["coo"][1], fresh1[v]["coo"][1]) for v in s1)
        s2 = jf(x2, pdict, unitdict, condition)
        assert s2 is not s1 and all(s2[v]["coo"][2] is s1[v]["coo"][2] and np.array_equal(s2[v]["coo"][2], fresh2[v]["coo"][2]) for v in s2)
