"""GPU parity tests of SURVEY.md 8f rows f-4 / f-2: the init / time / knot / terminal / time-ordering rows and the
device-form user constraint, through the C-ABI (gel_rows_*) and the reference-named Python functions, against the
golden fixtures of the imported reference (tests/golden/g11_knot_terminal.npz) and the numpy oracle.

Tolerances: linear rows bit-exact (differences of single variables plus a constant); node functions 1e-12 relative
(ocml vs libm); their forward-difference entries |d| <= 1e-5 + 1e-6 |ref| like every FD Jacobian entry of the path."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CONDS = {"Payload": {}, "Other_incl": {"OptimizationMode": "Other", "inclination": 42.3},
         "radius": {"altitude_perigee": None, "altitude_apogee": None}}


def example(extra=None):
    from gelato_amd import problem
    pdict, unitdict, condition, xdict = problem.make_problem("example")
    return pdict, unitdict, dict(condition, **(extra or {})), xdict


def xdict_of(E, x):
    return {k: np.ascontiguousarray(v) for k, v in E.split_x(x).items()}


@pytest.mark.parametrize("cname", list(CONDS))
def test_row_groups_vs_reference_golden_and_oracle(cname):
    from gelato_amd import con_dynamics
    from gelato_amd import con_init_terminal_knot as ck
    from gelato_amd import con_trajectory as ct
    from oracle import knot_terminal as kt
    g = load_golden("g11_knot_terminal.npz")
    pdict, unitdict, condition, _ = example(CONDS[cname])
    E = con_dynamics.engine_of(pdict, unitdict)
    sp = kt.make_spec(pdict, unitdict, condition)
    fns = {"init": (ck.equality_init, ck.equality_jac_init), "time": (ck.equality_time, ck.equality_jac_time),
           "knot": (ck.equality_knot_LGR, ck.equality_jac_knot_LGR),
           "terminal": (ck.equality_6DoF_LGR_terminal, ck.equality_jac_6DoF_LGR_terminal),
           "tineq": (ck.inequality_time, ck.inequality_jac_time),
           "rate": (ct.equality_6DoF_rate, ct.equality_jac_6DoF_rate), "imass": (ct.inequality_mass, ct.inequality_jac_mass),
           "kick": (ct.inequality_kickturn, ct.inequality_jac_kickturn)}
    for xname in ("init", "moved"):
        xd = xdict_of(E, g["x_" + xname])
        before = {k: v.copy() for k, v in xd.items()}
        for tag, (f, jf) in fns.items():
            base = "%s_%s_%s" % (xname, cname, tag)
            con, ref = f(xd, pdict, unitdict, condition), g[base + "_con"]
            J = jf(xd, pdict, unitdict, condition)
            if tag != "terminal":
                assert np.array_equal(np.asarray(con), ref), base
            else:
                assert np.all(np.abs(con - ref) <= 1e-13 + 1e-12 * np.abs(ref)), base
                assert np.all(np.abs(con - kt.equality_terminal(g["x_" + xname], sp)) <= 1e-13 + 1e-12 * np.abs(ref))
            for var, blk in J.items():
                k = base + "_jac_" + var
                assert np.array_equal(blk["coo"][0], g[k + "_rows"]) and np.array_equal(blk["coo"][1], g[k + "_cols"]), k
                assert blk["shape"] == tuple(g[k + "_shape"]) and blk["coo"][2].dtype == np.float64
                if tag != "terminal":
                    assert np.array_equal(blk["coo"][2], g[k + "_vals"]), k
                else:
                    d = np.abs(blk["coo"][2] - g[k + "_vals"])
                    assert np.all(d <= 1e-5 + 1e-6 * np.abs(g[k + "_vals"])), (k, d.max())
        assert all(np.array_equal(xd[k], before[k]) for k in xd)           # xdict is never modified
        assert con_dynamics.last_status(pdict) == 0


def test_device_form_user_constraint_vs_reference_jac_fd_golden():
    from gelato_amd import con_dynamics, con_user
    from gelato_amd.examples import user_constraints as uc
    g = load_golden("g11_knot_terminal.npz")
    pdict, unitdict, condition, _ = example()
    E = con_dynamics.engine_of(pdict, unitdict)
    con_user.set_user_module(uc)
    try:
        for xname in ("init", "moved"):
            xd = xdict_of(E, g["x_" + xname])
            val = con_user.equality_user(xd, pdict, unitdict, condition)
            assert np.ndim(val) == 0 and abs(val - g[xname + "_user_con"][0]) <= 1e-13      # a scalar, like the reference's
            assert con_user.inequality_user(xd, pdict, unitdict, condition) is None
            assert con_user.inequality_jac_user(xd, pdict, unitdict, condition) is None
            J = con_user.equality_jac_user(xd, pdict, unitdict, condition)
            assert sorted(J) == sorted(xd) and all(J[k].shape == (1, xd[k].size) for k in xd)   # dense, every key
            for key in xd:
                nz = np.nonzero(J[key][0])[0]
                assert np.array_equal(nz, g["%s_user_jac_%s_nzcols" % (xname, key)]), key
                ref = g["%s_user_jac_%s_nzvals" % (xname, key)]
                assert np.all(np.abs(J[key][0, nz] - ref) <= 1e-5 + 1e-6 * np.abs(ref)), key
        # a callable-form module keeps working (the reference's loop, column by column, on the user's Python)
        class Callable:
            @staticmethod
            def equality_user(xdict, pdict, unitdict, condition):
                return np.array([xdict["mass"][3] * 2.0 - xdict["t"][1]])

            @staticmethod
            def inequality_user(xdict, pdict, unitdict, condition):
                return None
        con_user.set_user_module(Callable)
        xd = xdict_of(E, g["x_init"])
        J = con_user.equality_jac_user(xd, pdict, unitdict, condition)
        assert abs(J["mass"][0, 3] - 2.0) < 1e-6 and abs(J["t"][0, 1] + 1.0) < 1e-6 and np.count_nonzero(J["position"]) == 0
    finally:
        con_user.set_user_module(None)


def test_rows_batch_device_api_and_nonfinite_status():
    import torch
    from gelato_amd import con_dynamics
    from gelato_amd import con_init_terminal_knot as ck
    from gelato_amd import problem
    pdict, unitdict, condition, xdict = example({"inclination": 40.0})
    pdict["gelato_amd_user_rows"] = (("periapsis_radius", "IIP_END", 6378137.0, 1.0), ("speed", "SEIG", 1000.0, 0.0))
    E = con_dynamics.engine_of(pdict, unitdict)
    R = ck.rows_of(pdict, unitdict, condition)
    from gelato_amd import pack_x
    x0 = pack_x(xdict)
    B = 37
    X = problem.synthetic_batch(x0, E.M, B, seed=5)
    con, jfn, rc = E.rows_eval(X)
    assert rc == 0 and con.shape == (B, R.nlin + R.nfn) and jfn.shape == (B, R.nfn, 7)
    for b in (0, 11, B - 1):                                 # element b of a batch == the single-vector call, bit for bit
        c1, j1, _ = E.rows_eval(X[b])
        assert np.array_equal(c1[0], con[b]) and np.array_equal(j1[0], jfn[b])
    dev = torch.device("cuda:0")
    dX = torch.from_numpy(X).to(dev)
    dcon = torch.empty((B, R.nlin + R.nfn), dtype=torch.float64, device=dev)
    djfn = torch.empty((B, R.nfn, 7), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    E.rows_eval_device(B, dX.data_ptr(), dcon.data_ptr(), djfn.data_ptr(), s)
    assert E.sync(s) == 0
    assert np.array_equal(dcon.cpu().numpy(), con) and np.array_equal(djfn.cpu().numpy(), jfn)
    # large host batch: the copy path instead of the zero-copy path, same bits
    Xl = np.tile(X, (8, 1))
    cl, jl, rc = E.rows_eval(Xl)
    assert rc == 0 and np.array_equal(cl[:B], con) and np.array_equal(jl[B:2 * B], jfn)
    # the speed row: |v| / 1000 of SEIG's first node; its position columns are exact zeros
    node = pdict["ps_params"].index_start_x(pdict["event_index"]["SEIG"])
    v = X[0, 4 * E.M + 3 * node:4 * E.M + 3 * node + 3] * unitdict["velocity"]
    k = R.waypoint_base - 1                                   # the last user row (the waypoint rows follow them)
    assert abs(con[0, R.nlin + k] - np.linalg.norm(v) / 1000.0) <= 1e-12 and np.all(jfn[0, k, 0:3] == 0.0)
    assert np.all(jfn[0, k, 6] == 0.0)                        # and it does not read the knot time
    # non-finite input -> status 1, reported through the sticky status of the callbacks
    xb = x0.copy()
    xb[E.M + 3 * (E.M - 1)] = np.nan
    _, _, rc = E.rows_eval(xb)
    assert rc == 1
    _, _, rc = E.rows_eval(x0)
    assert rc == 0


def test_callbacks_carry_every_row_group_and_a_sticky_status():
    from gelato_amd import con_dynamics, con_user, driver
    from gelato_amd.examples import user_constraints as uc
    pdict, unitdict, condition, xdict = example()
    con_user.set_user_module(uc)
    try:
        objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
        funcs, fail = objfunc(xdict)
        fs, fail2 = sens(xdict, funcs)
        assert not fail and not fail2
        for key in ("eqcon_init", "eqcon_time", "eqcon_knot", "eqcon_terminal", "eqcon_user", "ineqcon_time",
                    "eqcon_dyn_mass", "eqcon_dyn_pos", "eqcon_dyn_vel", "eqcon_dyn_quat"):
            assert funcs[key] is not None and fs[key] is not None, key
        assert funcs["ineqcon_user"] is None and fs["ineqcon_user"] is None
        assert funcs["eqcon_knot"].shape == (121,) and funcs["eqcon_terminal"].shape == (2,)
        # the shipped FlightConstraint: FAIRING altitude (exact) and impact-point longitude (min), one antenna at SECO
        assert funcs["eqcon_pos"].shape == (1,) and funcs["ineqcon_iip"].shape == (1,) and funcs["ineqcon_antenna"].shape == (1,)
        assert funcs["eqcon_iip"] is None and funcs["ineqcon_pos"] is None and fs["eqcon_iip"] is None and fs["ineqcon_pos"] is None
        assert sorted(fs["eqcon_pos"]) == ["position", "t"] and sorted(fs["ineqcon_iip"]) == ["position", "t", "velocity"]
        assert sorted(fs["ineqcon_antenna"]) == ["position", "t"]
        assert len(funcs) == 23 and sorted(funcs) == sorted(fs)          # every key of Trajectory_Optimization.py:194-312
        # the registration the reference does with pyoptsparse: every existing group with its wrt list; the Jacobian of a
        # group names exactly variables of that list (a block the list lacks would be dropped silently by addConGroup)
        groups = driver.constraint_groups(funcs, fs, condition)
        assert len(groups) == sum(v is not None for k, v in funcs.items() if k != "obj")
        for key, size, lo, up, wrt, jac in groups:
            assert set(jac) <= set(wrt), (key, sorted(jac), wrt)
            assert size == np.size(funcs[key]) and lo == 0.0 and (up is None) == key.startswith("ineqcon")
            for var, blk in jac.items():
                shape = blk["shape"] if isinstance(blk, dict) else blk.shape
                assert shape[0] == size, (key, var, shape)
        # a NaN that only the terminal rows see (velocity of the very last node feeds no defect row's RHS ... but does
        # feed D.X of the last phase): whatever group sees it first, the callback reports it once, at the end
        bad = {k: v.copy() for k, v in xdict.items()}
        bad["velocity"][-1] = np.nan
        _, fail = objfunc(bad)
        assert fail
        _, fail = objfunc(xdict)                                  # and the flag does not stick to the next x
        assert not fail
        assert con_dynamics.last_status(pdict) == 0
    finally:
        con_user.set_user_module(None)


def test_user_rows_registered_after_the_state_exists():
    """The state (engine, row table) of a pdict created BEFORE the user module is known -- a direct con_* call or engine_of()
    ahead of make_callbacks -- must not make the first pinned callback hand equality_user the rows of another group: the user
    rows are registered before the callback's first look at the row table, and a table pinned for one tuple of user rows is not
    handed out for another."""
    from gelato_amd import con_dynamics, con_user, driver
    from gelato_amd import con_init_terminal_knot as ck
    from gelato_amd.examples import user_constraints as uc
    pdict, unitdict, condition, xdict = example()
    con_user.set_user_module(None)
    try:
        con_dynamics.engine_of(pdict, unitdict)                       # the state exists ...
        ck.equality_init(xdict, pdict, unitdict, condition)           # ... and so does a row table WITHOUT user rows
        assert ck.rows_of(pdict, unitdict, condition).user_rows == []
        con_user.set_user_module(uc)                                  # only now the user module is set
        objfunc, sens = driver.make_callbacks(pdict, unitdict, condition)
        funcs, fail = objfunc(xdict)
        assert not fail
        # the reference value of the shipped user constraint (periapsis radius at IIP_END), from a fresh problem
        p2, u2, c2, x2 = example()
        want = con_user.equality_user(x2, p2, u2, c2)
        assert np.array_equal(np.asarray(funcs["eqcon_user"]), np.asarray(want))
        assert len(ck.rows_of(pdict, unitdict, condition).user_rows) == 1
        fs, fail = sens(xdict, funcs)
        assert not fail and fs["eqcon_user"]["position"].shape[0] == 1
        # switching the module between runs on the same pdict
        con_user.set_user_module(None)
        funcs, _ = objfunc(xdict)
        assert funcs["eqcon_user"] is None
        # an exception inside a callback does not leave xdict pinned
        bad = dict(xdict)
        del bad["u"]
        with pytest.raises(KeyError):
            objfunc(bad)
        assert pdict["_gelato_amd"]._pinned is None
    finally:
        con_user.set_user_module(None)


def test_one_round_trip_callback_equals_the_separate_calls():
    """gel_eval_callback: defect groups + row table + aero kinds of one decision vector, launched back to back, one
    synchronise -- every output bit for bit what the separate entry points return; configuring a kind or the row
    table afterwards invalidates the shared frame of the Python mirrors."""
    from gelato_amd import con_aero, con_dynamics, pack_x
    from gelato_amd import con_init_terminal_knot as ck
    pdict, unitdict, condition, xdict = example({"inclination": 51.6})
    cond = dict(condition, AOA_max={"KICKTURN": {"value": 10.0, "range": "all"}}, dynamic_pressure_max={},
                Q_alpha_max={"ZEROLIFT_START": {"value": 30000.0, "range": "all"}, "ZEROLIFT_END": {"value": 2.0e4, "range": "initial"}})
    E = con_dynamics.engine_of(pdict, unitdict)
    R = ck.rows_of(pdict, unitdict, cond)
    for kind in ("alpha", "q", "qalpha"):
        con_aero._configured(pdict, unitdict, cond, kind)
    x = pack_x(xdict) * (1.0 + 1e-6)
    for want_jac in (False, True):
        fr = E.eval_callback(x, want_jac)
        assert fr["rc"] == 0
        res, vals, _ = E.eval(x)
        assert np.array_equal(fr["res"], res)
        rc_, rj_, _ = E.rows_eval(x, want_jac=True)
        assert np.array_equal(fr["rows_con"], rc_[0])
        ac, aj, _ = E.eval_aero_all(x, want_jac=True)
        assert sorted(fr["aero_con"]) == ["alpha", "qalpha"]
        for kind in fr["aero_con"]:
            assert np.array_equal(fr["aero_con"][kind], ac[kind][0])
        if want_jac:
            assert np.array_equal(fr["vals"], vals) and np.array_equal(fr["rows_jfn"], rj_[0])
            for kind in fr["aero_jac"]:
                assert np.array_equal(fr["aero_jac"][kind], aj[kind][0])
        else:
            assert fr["vals"] is None and fr["rows_jfn"] is None and not fr["aero_jac"]
    # through the reference-named functions: one frame per xdict, shared by the three modules
    xd = {k: np.ascontiguousarray(v) for k, v in E.split_x(x).items()}
    st = con_dynamics._state(pdict, unitdict)
    a = con_dynamics.equality_dynamics_velocity(xd, pdict, unitdict, cond)
    f1 = st._frame
    b = ck.equality_knot_LGR(xd, pdict, unitdict, cond)
    c = con_aero.inequality_max_qalpha(xd, pdict, unitdict, cond)
    assert st._frame is f1 and not f1["jac"]                       # values: one evaluation for all three modules
    assert np.array_equal(a, E.split_res(res)["vel"]) and np.array_equal(b, rc_[0][slice(*R.slices["knot"])])
    assert np.array_equal(c, ac["qalpha"][0])
    J = ck.equality_jac_6DoF_LGR_terminal(xd, pdict, unitdict, cond)
    f2 = st._frame
    assert f2 is not f1 and f2["jac"]                              # derivatives asked: one more evaluation, with them
    con_dynamics.equality_jac_dynamics_quaternion(xd, pdict, unitdict, cond)
    con_aero.inequality_jac_max_alpha(xd, pdict, unitdict, cond)
    assert st._frame is f2 and J["position"]["coo"][2].shape == (9,)
    # a reconfiguration invalidates the frame (same xdict, new limits)
    cond2 = dict(cond, AOA_max={"KICKTURN": {"value": 5.0, "range": "all"}})
    d = con_aero.inequality_max_alpha(xd, pdict, unitdict, cond2)
    assert st._frame is not f2
    assert np.allclose(1.0 - d, 2.0 * (1.0 - con_aero.inequality_max_alpha(xd, pdict, unitdict, cond)), rtol=1e-12)
