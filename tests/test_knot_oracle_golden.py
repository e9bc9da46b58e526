"""CPU tests of SURVEY.md 8f rows f-4 / f-2 / f-3: the numpy oracle (oracle/knot_terminal.py) against the golden fixtures
written from the imported reference (lib/con_init_terminal_knot.py, example/user_constraints.py + lib/jac_fd.py,
initialize.py), and the host logic of the product (row tables, constant COO blocks, gel_initial_guess) against both.
No GPU: the product side uses host-only handles and never evaluates."""
import numpy as np
import pytest

from conftest import D_tau_from_golden, load_golden, problem_from_golden
from gelato_amd import Engine, problem
from oracle import knot_terminal as kt

CONDS = {"Payload": {}, "Other_incl": {"OptimizationMode": "Other", "inclination": 42.3},
         "radius": {"altitude_perigee": None, "altitude_apogee": None}}
GROUPS = [("init", kt.equality_init, kt.equality_jac_init), ("time", kt.equality_time, kt.equality_jac_time),
          ("knot", kt.equality_knot_LGR, kt.equality_jac_knot_LGR),
          ("terminal", kt.equality_terminal, kt.equality_jac_terminal),
          ("tineq", kt.inequality_time, kt.inequality_jac_time),
          ("rate", kt.equality_rate, kt.equality_jac_rate), ("imass", kt.inequality_mass, kt.inequality_jac_mass),
          ("kick", kt.inequality_kickturn, kt.inequality_jac_kickturn)]


def example(extra=None):
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    pdict["device"] = -1                                  # host-only handle: describes, never evaluates
    return pdict, unitdict, dict(condition, **(extra or {})), xdict


@pytest.mark.parametrize("xname", ["init", "moved"])
@pytest.mark.parametrize("cname", list(CONDS))
def test_oracle_rows_vs_reference_golden(xname, cname):
    g = load_golden("g11_knot_terminal.npz")
    pdict, unitdict, condition, _ = example(CONDS[cname])
    sp = kt.make_spec(pdict, unitdict, condition)
    assert sp["time_ref"] == list(g["time_ref_index"]) and np.array_equal(sp["mass_jettison"], g["mass_jettison"])
    x = g["x_" + xname]
    for tag, f, jf in GROUPS:
        base = "%s_%s_%s" % (xname, cname, tag)
        con, ref = f(x, sp), g[base + "_con"]
        assert con.shape == ref.shape and np.all(np.abs(con - ref) <= 1e-13 + 1e-12 * np.abs(ref)), base
        for var, blk in jf(x, sp).items():
            b = base + "_jac_" + var
            assert np.array_equal(blk["coo"][0], g[b + "_rows"]) and np.array_equal(blk["coo"][1], g[b + "_cols"]), b
            assert blk["coo"][0].dtype == np.int32 and blk["shape"] == tuple(g[b + "_shape"]), b
            if tag == "terminal":   # forward differences of the same arithmetic: FD noise only
                assert np.all(np.abs(blk["coo"][2] - g[b + "_vals"]) <= 1e-6 + 1e-6 * np.abs(g[b + "_vals"])), b
            else:
                assert np.array_equal(blk["coo"][2], g[b + "_vals"]), b


def test_oracle_orbital_functions_and_user_constraint_vs_golden():
    g = load_golden("g11_knot_terminal.npz")
    r, v = g["orb_pos"], g["orb_vel"]
    el = np.array([kt.orbital_elements(a, b) for a, b in zip(r, v)])
    assert np.all(np.abs(el[:, :2] - g["orb_elements"][:, :2]) <= 1e-12 * np.abs(g["orb_elements"][:, :2]))
    assert np.all(np.abs(el[:, 2:] - g["orb_elements"][:, 2:]) <= 1e-7)           # degrees; acos near 0 is ill-conditioned
    for name, f in (("orb_angmom", kt.angular_momentum), ("orb_energy", kt.orbit_energy), ("orb_incl", kt.inclination_rad)):
        out = np.array([f(a, b) for a, b in zip(r, v)])
        assert np.all(np.abs(out - g[name]) <= 1e-13 * np.abs(g[name])), name
    assert np.allclose([kt.angular_momentum_from_altitude(2.0e5, 3.0e5), kt.orbit_energy_from_altitude(2.0e5, 3.0e5)],
                       g["orb_from_alt"], rtol=1e-15, atol=0)
    pdict, unitdict, condition, _ = example()
    sp = kt.make_spec(pdict, unitdict, condition)
    sec = int(g["user_knot_index"])
    M = sp["M"]
    for xname in ("init", "moved"):
        x = g["x_" + xname]
        f = lambda y: kt.user_apogee_height(y, sp, sec)   # noqa: E731
        assert abs(f(x) - g[xname + "_user_con"][0]) <= 1e-13
        # the generic loop of lib/jac_fd.py over the columns the constraint can see (+ a few it cannot)
        J = np.zeros(x.size)
        node = sp["xa"][sec]
        cols = [M + 3 * node + c for c in range(3)] + [4 * M + 3 * node + c for c in range(3)] + [0, M, 7 * M + 5, x.size - 1]
        g0 = f(x)
        for c in cols:
            xp = x.copy(); xp[c] += sp["dx"]
            J[c] = (f(xp) - g0) / sp["dx"]
        assert np.array_equal(np.nonzero(J[M:4 * M])[0], g[xname + "_user_jac_position_nzcols"])
        assert np.array_equal(np.nonzero(J[4 * M:7 * M])[0], g[xname + "_user_jac_velocity_nzcols"])
        for key, off in (("position", M), ("velocity", 4 * M)):
            ref = g["%s_user_jac_%s_nzvals" % (xname, key)]
            got = J[off + g["%s_user_jac_%s_nzcols" % (xname, key)]]
            assert np.all(np.abs(got - ref) <= 1e-6 + 1e-6 * np.abs(ref)), key
        for key in ("mass", "quaternion", "u", "t"):
            assert g["%s_user_jac_%s_nzcols" % (xname, key)].size == 0


def test_oracle_initial_guess_vs_reference_output():
    g, g6 = load_golden("g12_initial_guess.npz"), load_golden("g6_example.npz")
    v = problem.load_vehicle()
    tab = np.array(v["trajectory"])
    col = {c: i for i, c in enumerate(v["trajectory_columns"])}
    table = tab[:, [col[c] for c in problem._TRAJ_COLS]]
    nodes = [int(n) for n in g["num_nodes"]]
    prob = problem_from_golden(g6)
    units = dict(zip(("mass", "position", "velocity", "u", "t"), prob["units"]))
    out = kt.initial_guess(tab[:, col["time"]], table, g["knot_times"], nodes, [g["tau_%d" % n] for n in nodes], units)
    for k, val in out.items():
        assert np.array_equal(val, g["example_" + k]), k                  # the reference's own bits
    # the product's host entry point (gel_initial_guess) on a host-only handle, same tau: same bits
    D, tau = D_tau_from_golden(g6, prob)
    E = Engine(prob, D=D, tau=tau, device=-1)
    X = E.split_x(E.initial_guess(tab[:, col["time"]], table, g["knot_times"]))
    for k in X:
        assert np.array_equal(X[k], g["example_" + k]), k
    with pytest.raises(Exception, match="must not decrease"):
        E.initial_guess(tab[::-1, col["time"]], table, g["knot_times"])
    # a reference table that ends in a repeated time, with knots beyond it: the extrapolation interval has zero width; the
    # non-finite guess comes back like the reference's would, and the caller is told (GEL_NONFINITE -> RuntimeWarning)
    t2 = tab[:, col["time"]].copy()
    t2[-1] = t2[-2]
    knots = g["knot_times"].copy()
    knots[-1] = t2[-1] + 50.0
    with pytest.warns(RuntimeWarning, match="non-finite"):
        xbad = E.initial_guess(t2, table, knots)
    assert not np.all(np.isfinite(xbad))


@pytest.mark.parametrize("cname", list(CONDS))
def test_product_row_tables_and_constant_jacobians_vs_golden(cname):
    """Host logic of gelato_amd.con_init_terminal_knot without a GPU: the linear rows reproduce the reference's values
    bit for bit when evaluated as the kernel does ((c0 x0 + c1 x1) + c), and the constant COO blocks are the reference's."""
    from gelato_amd import con_init_terminal_knot as ck
    g = load_golden("g11_knot_terminal.npz")
    pdict, unitdict, condition, xdict = example(CONDS[cname])
    R = ck.rows_of(pdict, unitdict, condition)
    from gelato_amd import con_trajectory as ct
    jfs = {"init": ck.equality_jac_init, "time": ck.equality_jac_time, "knot": ck.equality_jac_knot_LGR,
           "tineq": ck.inequality_jac_time, "rate": ct.equality_jac_6DoF_rate, "imass": ct.inequality_jac_mass,
           "kick": ct.inequality_jac_kickturn}
    for xname in ("init", "moved"):
        x = g["x_" + xname]
        lin = np.array([(c0 * x[i0] + (c1 * x[i1] if i1 >= 0 else 0.0)) + cc if i1 >= 0 else c0 * x[i0] + cc
                        for (i0, c0, i1, c1, cc) in R.lin])
        for tag, jf in jfs.items():
            base = "%s_%s_%s" % (xname, cname, tag)
            a, b = R.slices[tag]
            assert np.array_equal(lin[a:b], g[base + "_con"]), base
            J = jf(xdict, pdict, unitdict, condition)
            ref_vars = sorted(k[len(base) + 5:-5] for k in g if k.startswith(base + "_jac_") and k.endswith("_rows"))
            assert sorted(J) == ref_vars, (base, sorted(J), ref_vars)
            for var, blk in J.items():
                k = base + "_jac_" + var
                assert all(np.array_equal(blk["coo"][i], g[k + s]) for i, s in enumerate(("_rows", "_cols", "_vals"))), k
                assert blk["coo"][0].dtype == np.int32 and blk["shape"] == tuple(g[k + "_shape"])
    # terminal rows: functions, node and scaling of the node-function rows
    assert ct.equality_length_6DoF_rate(xdict, pdict, unitdict, condition) == len(g["init_%s_rate_con" % cname]) == 93
    nT = 3 if cname == "Other_incl" else 2
    assert R.n_terminal == nT and [f[0] for f in R.fn[:nT]] == ["orbit_energy", "angular_momentum", "inclination_rad"][:nT]
    assert all(f[1] == pdict["M"] - 1 for f in R.fn[:nT])
    sp = kt.make_spec(pdict, unitdict, condition)
    c_t, e_t = kt._terminal_targets(sp)
    assert R.fn[0][2] == e_t and R.fn[1][2] == c_t
    with pytest.raises(Exception, match="host-only"):
        ck.equality_init(xdict, pdict, unitdict, condition)


def test_device_form_user_module_declares_the_shipped_constraint():
    from gelato_amd.examples import user_constraints as uc
    from gelato_amd.usercon_tools import NodeFunction, get_index_event, get_value
    assert list(uc.EQUALITY_ROWS) == [("periapsis_radius", "IIP_END", 6378137.0, 1.0)] and not uc.INEQUALITY_ROWS
    with pytest.raises(ValueError):
        NodeFunction("no_such_function", "IIP_END")
    pdict, unitdict, condition, xdict = example()
    i = pdict["event_index"]["IIP_END"]
    xa = pdict["ps_params"].index_start_x(i)
    assert get_index_event(pdict, "IIP_END", "position") == (3 * xa, 3 * (xa + pdict["ps_params"].nodes(i) + 1))
    assert np.array_equal(get_value(xdict, pdict, unitdict, "IIP_END", "velocity"),
                          xdict["velocity"][3 * xa:3 * xa + 3] * unitdict["velocity"])


def test_inequality_mass_raises_like_the_reference_when_a_stage_event_is_missing():
    """lib/con_trajectory.py:40-49 indexes the empty list of sections named like the stage's ignition_at / cutoff_at event:
    IndexError.  Only the synthetic bench meshes (cut-down event lists, flagged by problem.make_problem) skip such a stage."""
    import warnings
    from gelato_amd import con_init_terminal_knot as ck2
    from gelato_amd import problem as pb
    pdict, unitdict, condition, _ = pb.make_problem("example")
    pdict["device"] = -1
    pdict["RocketStage"] = {k: dict(v) for k, v in pdict["RocketStage"].items()}
    first = next(iter(pdict["RocketStage"].values()))
    first["cutoff_at"] = "NO_SUCH_EVENT"
    # raised where the reference raises -- inside inequality_mass / inequality_jac_mass -- and nowhere else: the shared row
    # table is built, and every other group of it (knot, terminal, time ...) works on such a pdict
    from gelato_amd import con_trajectory as ct
    T = ck2.rows_of(pdict, unitdict, condition)
    assert T.missing_stage_events == (first["ignition_at"], "NO_SUCH_EVENT")
    assert T.slices["knot"][1] > T.slices["knot"][0] and "t" in T.jac["tineq"]
    for fn in (ct.inequality_mass, ct.inequality_jac_mass):
        with pytest.raises(IndexError, match="NO_SUCH_EVENT"):
            fn({}, pdict, unitdict, condition)
    assert ct.equality_length_6DoF_rate({}, pdict, unitdict, condition) > 0
    assert pb.make_problem("mixed-6x64")[0]["gelato_amd_allow_missing_stage_events"]
    pdict["gelato_amd_allow_missing_stage_events"] = True
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ck2.rows_of(pdict, unitdict, dict(condition)).missing_stage_events is None     # a new condition object: the table is rebuilt
    assert any("inequality_mass" in str(x.message) for x in w)
