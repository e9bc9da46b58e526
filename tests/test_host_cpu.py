"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/gelato_amd.h declares, the C++ LGR generator matches the reference's goldens, the Python
mirror of the reference interface has the right names/signatures, the problem builder reproduces
the fixture inputs, and the engine refuses to run without a GPU (no CPU fallback)."""
import inspect
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, problem_from_golden

import gelato_amd
from gelato_amd import _lib, con_dynamics, cost_gradient, problem
from gelato_amd.PSfunctions import differentiation_matrix_LGR, nodes_LGR
from gelato_amd.SectionParameters import PSparams


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gelato_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gel_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(L, s), "libgelato_amd.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert b"gfx950" in L.gel_version()


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 8, 16, 32, 64, 128])
def test_lgr_generator_matches_reference(n):
    g = load_golden("g1_lgr.npz")
    tau, D = nodes_LGR(n), differentiation_matrix_LGR(n)
    assert D.shape == (n, n + 1)
    assert np.max(np.abs(tau - g["tau_%d" % n])) <= 1e-14
    Dg = g["D_%d" % n]
    rowmax = np.max(np.abs(Dg), axis=1, keepdims=True)
    assert np.max(np.abs(D - Dg) / rowmax) <= 1e-11     # SURVEY.md 8c tolerance for D
    assert np.max(np.abs(D.sum(axis=1))) <= 1e-12 * np.abs(D).max()
    # D differentiates polynomials of degree <= n exactly: D @ tau_x = 1, D @ tau_x^2 = 2 tau
    tx = np.concatenate([[-1.0], tau])
    assert np.max(np.abs(D @ tx - 1.0)) <= 1e-11 * np.abs(D).max()
    assert np.max(np.abs(D @ tx ** 2 - 2 * tau)) <= 1e-11 * np.abs(D).max()


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 16, 32, 64])
def test_lgr_unflipped_set_matches_reference(n):
    """nodes_LGR / differentiation_matrix_LGR with reverse=False (lib/PSfunctions.py:149-168,182-208): the set that contains -1"""
    g = load_golden("g1b_lgr_unflipped.npz")
    tau, D = nodes_LGR(n, reverse=False), differentiation_matrix_LGR(n, reverse=False)
    assert tau[0] == -1.0 and np.all(np.diff(tau) > 0) and D.shape == (n, n + 1) and D.flags.c_contiguous
    assert np.max(np.abs(tau - g["tau_%d" % n])) <= 1e-14
    Dg = g["D_%d" % n]
    assert np.max(np.abs(D - Dg) / np.max(np.abs(Dg), axis=1, keepdims=True)) <= 1e-11
    tx = np.concatenate([tau, [1.0]])
    assert np.max(np.abs(D @ tx ** 2 - 2 * tau)) <= 1e-11 * np.abs(D).max()


def test_lgr_rejects_small_n():
    with pytest.raises(_lib.GelatoAmdError):
        nodes_LGR(1)


def test_psparams_interface():
    ps = PSparams([5, 5, 16, 8, 2])
    assert ps.num_sections() == 5 and ps.num_u() == 36 and ps.num_x() == 41
    assert ps.get_index(0) == (0, 5, 0, 6, 5)
    assert ps.get_index(2) == (10, 26, 12, 29, 16)      # SectionParameters.py:97-103
    assert ps.index_start_x(3) == 29 and ps.index_end_x(3) == 38 and ps.index_end_u(3) == 34
    t = ps.time_nodes(2, 0.25, 0.75)
    assert t.shape == (17,) and t[0] == 0.25 and abs(t[-1] - 0.75) < 1e-15
    assert ps[1]["nodes"] == 5 and ps[1]["D"].shape == (5, 6)
    with pytest.raises(ValueError):
        ps.D(5)
    with pytest.raises(ValueError):
        ps.tau(-1)


def test_shim_mirrors_reference_signatures():
    names = ["equality_dynamics_mass", "equality_jac_dynamics_mass", "equality_dynamics_position",
             "equality_jac_dynamics_position", "equality_dynamics_velocity", "equality_jac_dynamics_velocity",
             "equality_dynamics_quaternion", "equality_jac_dynamics_quaternion"]
    for n in names:
        fn = getattr(con_dynamics, n)
        assert list(inspect.signature(fn).parameters) == ["xdict", "pdict", "unitdict", "condition"]
    from gelato_amd import dynamics, jac_fd
    assert list(inspect.signature(dynamics.dynamics_velocity).parameters)[:9] == [
        "mass_e", "pos_eci_e", "vel_eci_e", "quat_eci2body", "t", "param", "wind_table", "CA_table", "units"]
    assert list(inspect.signature(dynamics.dynamics_quaternion).parameters) == ["quat_eci2body", "u_e", "unit_u"]
    assert list(inspect.signature(jac_fd.jac_fd).parameters) == ["con", "xdict", "pdict", "unitdict", "condition"]


def test_cost_functions():
    g6 = load_golden("g6_example.npz")
    g8 = load_golden("g8_cost.npz")
    M = int(problem_from_golden(g6)["num_nodes"].sum()) + len(problem_from_golden(g6)["num_nodes"])
    x = g6["x"]
    xd = {"mass": x[:M], "t": x[-13:]}
    for mode, key in [("Payload", "mass"), ("Other", "t")]:
        cond = {"OptimizationMode": mode}
        assert cost_gradient.cost_6DoF(xd, cond) == float(g8["cost_" + mode])
        jac = cost_gradient.cost_jac(xd, cond)
        assert list(jac) == [key] and np.array_equal(jac[key], g8["costjac_%s_%s" % (mode, key)])


@pytest.mark.parametrize("name,gname", [("example", "example"), ("3x32", "3x32"), ("mixed-6x64", "mixed6x64")])
def test_problem_builder_reproduces_fixture_inputs(name, gname):
    g = load_golden("g6_%s.npz" % gname)
    prob_ref = problem_from_golden(g)
    pdict, unitdict, condition, xdict = problem.make_problem(name)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    for k in ["num_nodes", "thrust", "reference_area", "nozzle_area", "engine_on", "attitude_hold", "units"]:
        assert np.array_equal(prob[k], prob_ref[k]), k
    assert np.allclose(prob["massflow"], prob_ref["massflow"], rtol=1e-15, atol=0)
    assert np.allclose(prob["wind_table"], prob_ref["wind_table"], rtol=0, atol=1e-12)
    assert np.array_equal(prob["ca_table"], prob_ref["ca_table"])
    x = gelato_amd.pack_x(xdict)
    assert x.shape == g["x"].shape
    assert np.max(np.abs(x - g["x"])) <= 1e-12           # tau differs by <= 1e-14, interpolation is linear


def test_synthetic_batch_is_deterministic_and_outward():
    pdict, unitdict, _, xdict = problem.make_problem("3x32")
    x0 = gelato_amd.pack_x(xdict)
    M = pdict["M"]
    X = problem.synthetic_batch(x0, M, 4)
    assert np.array_equal(X[0], x0)
    assert np.array_equal(X, problem.synthetic_batch(x0, M, 4))
    pos0 = np.linalg.norm(x0[M:4 * M].reshape(-1, 3), axis=1)
    for b in range(1, 4):
        assert np.all(np.linalg.norm(X[b, M:4 * M].reshape(-1, 3), axis=1) >= pos0)
        assert np.max(np.abs(X[b] / x0 - 1.0)[x0 != 0]) < 1e-5


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pdict, unitdict, _, xdict = problem.make_problem("3x32")
    with pytest.raises(_lib.GelatoAmdError, match="no HIP device|HIP"):
        con_dynamics.equality_dynamics_mass(xdict, pdict, unitdict, None)
    from gelato_amd import dynamics
    with pytest.raises(_lib.GelatoAmdError):
        dynamics.dynamics_quaternion(np.zeros((2, 4)), np.zeros((2, 2)), 1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gelato_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gelato_oracle" not in txt, f


def test_jac_fd_of_a_user_function_matches_reference_loop(tmp_path, monkeypatch):
    """SURVEY f-2: lib/jac_fd.py + lib/con_user.py on a user's own Python function (golden G10 = the reference's
    loop on tests/golden/user_function.py).  No engine is involved: host logic only."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from user_function import scalar_con, user_con
    from gelato_amd import jac_fd
    g = load_golden("g10_jacfd_user.npz")
    g6 = load_golden("g6_example.npz")
    nn = problem_from_golden(g6)["num_nodes"]
    N, S = int(nn.sum()), len(nn)
    M = N + S
    x = g["x"]
    o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N, S + 1])
    keys = ["mass", "position", "velocity", "quaternion", "u", "t"]
    xd = {k: x[o[i]:o[i + 1]].copy() for i, k in enumerate(keys)}
    keep = {k: v.copy() for k, v in xd.items()}
    pdict = {"dx": float(g["dx"])}
    unitdict = {"mass": 27442.0, "position": 6378137.0, "velocity": 1000.0, "u": 1.0, "t": 597.0}
    for name, fn in (("user", user_con), ("scalar", scalar_con)):
        J = jac_fd.jac_fd(fn, xd, pdict, unitdict, None)
        assert list(J) == keys                                       # every key, reference order
        for k in keys:
            ref = g["%s_%s" % (name, k)]
            assert J[k].shape == ref.shape
            # the reference's in-place += / -= leaves <= 1.1e-16 of drift in x (golden 'drift'), i.e. <= ~1e-8 here
            assert np.all(np.abs(J[k] - ref) <= 1e-6 + 1e-7 * np.abs(ref)), (name, k, np.abs(J[k] - ref).max())
    assert all(np.array_equal(xd[k], keep[k]) for k in keys)          # the caller's arrays are never touched
    # con_user: no user_constraints module -> behaves like _user_constraints_empty.py
    from gelato_amd import con_user
    monkeypatch.setattr(con_user, "_mod", None)
    assert con_user.equality_user(xd, pdict, unitdict, None) is None
    assert con_user.equality_jac_user(xd, pdict, unitdict, None) is None
    # ... and with one on the path it is differentiated by jac_fd (lib/con_user.py:33-42)
    (tmp_path / "user_constraints.py").write_text(
        "from user_function import user_con as equality_user\n"
        "def inequality_user(xdict, pdict, unitdict, condition):\n    return None\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    monkeypatch.setattr(con_user, "_mod", None)
    sys.modules.pop("user_constraints", None)
    Ju = con_user.equality_jac_user(xd, pdict, unitdict, None)
    assert np.allclose(Ju["velocity"], g["user_velocity"], rtol=1e-7, atol=1e-6)
    assert con_user.inequality_jac_user(xd, pdict, unitdict, None) is None
    sys.modules.pop("user_constraints", None)
    monkeypatch.setattr(con_user, "_mod", None)


def test_static_counter_files_name_their_build():
    """profiles/traffic_*.json / fp64_*.json (what bench.py reads into roofline.traffic / roofline.fp64) carry the sha256 of the
    library they were recorded with, and the four BASELINE workloads are on record for ONE build: bench.py reports them only for
    that library (`null` + the reason otherwise)."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "fp64_*.json")))
    assert len(files) == 8
    shas = set()
    for f in files:
        d = json.load(open(f))
        assert isinstance(d.get("build_so_sha256"), str) and len(d["build_so_sha256"]) == 64, f
        assert "workload" in d and "batch" in d, f
        shas.add(d["build_so_sha256"])
    assert len(shas) == 1
    # ... and the hash of the gfx950 code objects inside it (what the counters describe): a host-only rebuild keeps them valid
    dev = {json.load(open(f)).get("build_device_code_sha256") for f in files}
    assert len(dev) == 1 and (None in dev or all(len(v) == 64 for v in dev))
    if _lib.so_sha256() in shas and None not in dev:
        assert _lib.device_code_sha256() in dev


def test_a_replaced_shard_plan_is_refused_not_used():
    """gel_shard_plan is state of the handle; a second plan on the same Engine makes the first UnitShards stale.  The
    stale object refuses to size a buffer or run a step (host-only handle: no GPU needed for the plan)."""
    from gelato_amd import Engine, parallel
    pdict, unitdict, _c, _x = problem.make_problem("mixed-6x64")
    E = Engine(con_dynamics.problem_arrays(pdict, unitdict), device=-1)
    sh2 = parallel.UnitShards(E, 2, 0)
    assert sh2.plan == (2, sh2.width) and E.shard_plan_key[:2] == sh2.plan
    sh2.check_current()
    sh8 = parallel.UnitShards(E, 8, 3)
    assert sh8.width != sh2.width
    sh8.check_current()
    with pytest.raises(RuntimeError, match="replaced"):
        sh2.check_current()
    with pytest.raises(RuntimeError, match="replaced"):
        sh2.buffer(4)
    # the same plan again is the same key: an object built from it is current, the older twin as well
    sh2b = parallel.UnitShards(E, 2, 1)
    sh2b.check_current(); sh2.check_current()
    with pytest.raises(RuntimeError, match="replaced"):
        sh8.check_current()


@pytest.mark.parametrize("name", ["example", "mixed-6x64", "stress-12x128"])
def test_jac_fd_blocks_cover_the_reference_pattern(name):
    """gel_jac_fd_blocks leaves out only exact zeros: every (row, column) of the reference's own sparsity pattern of a group
    (lib/con_dynamics.py:75-76,108-113) lies inside the block of the row's phase; blocks tile the group's rows; a phase's columns are
    its own state nodes, controls and knot times."""
    from gelato_amd.engine import BLOCKS, GROUPS
    pdict, unitdict, _c, xdict = problem.make_problem(name)
    pdict["device"] = -1
    E = con_dynamics.engine_of(pdict, unitdict)
    voff = {"mass": 0, "position": E.M, "velocity": 4 * E.M, "quaternion": 7 * E.M, "u": 11 * E.M, "t": 11 * E.M + 2 * E.N}
    pat = E.pattern()
    for gi, grp in enumerate(GROUPS):
        rows, cols, row0, off, cmap = E.jac_fd_block_dims(grp)
        assert row0[0] == 0 and np.array_equal(row0[1:], np.cumsum(rows)[:-1]) and int(rows.sum()) == E.nrows[gi]
        assert np.array_equal(off, np.concatenate([[0], np.cumsum(rows * cols)]))
        n = np.asarray(E.num_nodes)
        assert np.array_equal(cols, 13 * n + 13) and all(len(set(c.tolist())) == len(c) and c.max() < E.nvars for c in cmap)
        phase_of_row = np.repeat(np.arange(E.S), rows)
        inside = [set(c.tolist()) for c in cmap]
        for b, (g, var) in enumerate(BLOCKS):
            if g != grp:
                continue
            r, c = pat[b]
            assert all((int(ci) + voff[var]) in inside[phase_of_row[int(ri)]] for ri, ci in zip(r[::7], c[::7]))
    assert int(off[-1]) * 2 < E.nrows[3] * E.nvars                   # equal phases: 1/S of the dense matrix (0.167 at 6 x 64)
