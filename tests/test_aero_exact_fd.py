"""The gradients of the aero path constraints against EXACT finite-difference quotients
(tests/golden/g18_aero_exact_fd.npz: src/wrapper_utils.hpp:89-111,163-175 in 40-digit arithmetic on the fp64 inputs the sweeps of
lib/con_aero.py:311-371 form, tests/golden/make_aero_exact_fd.py).

Rounds 1-2 compared these gradients with a flat |d| <= 5e-4: alpha = acos(c) with c -> 1 makes the reference's own entries noisy
(eps / sin(alpha) / dx per evaluation) and a flat number either hides real errors at large angles or fails at small ones.  Here
every implementation is held to the bound that follows from the arithmetic, per entry (tests/fd_noise.py aero_bound):

  the reference itself (G9's values) and the oracle   |entry - exact| <= bound + drift   (drift: the in-place `+= dx, -= dx`)
  engine (default: t0 / tf columns as zeros)           |entry - exact| <= bound,  t columns == 0 exactly (the exact value)
  engine, GEL_FLAG_FD_RECOMPUTE                        |entry - exact| <= bound

CPU part: the oracle and the reference's values.  GPU part (-m gpu): the engine through the C-ABI."""
import numpy as np
import pytest

import fd_noise
import states
from conftest import D_tau_from_golden, load_golden, problem_from_golden
from test_aero_oracle_golden import KINDS, VARS, spec_from_golden

COLS = {"position": slice(0, 3), "velocity": slice(3, 6), "quaternion": slice(6, 10), "t": slice(10, 12)}
LIMITS = {"alpha": 0.2, "q": 4.0e4, "qalpha": 5.0e3}
# rows whose derived bound is not finite (the air-relative speed may vanish inside the wind table's range: the vertical ascent of
# mixed-6x64's first phase; five clamped rows of the synthetic G9 set), per fixture and kind: a regression cannot hide more rows there
UNBOUNDED_ROWS = {("g9_synthetic", "qalpha"): 5, ("mixed-6x64", "alpha"): 59, ("mixed-6x64", "q"): 59, ("mixed-6x64", "qalpha"): 59}
VALUE_BOUND_CAP = 1e-8       # on f = alpha / limit etc. (O(1) quantities): no row's value check is looser than this
CASES = ["g9_example", "g9_synthetic", "ragged", "polar", "layers", "breaks", "mixed-6x64", "stress-12x128"]
BASELINE = {"mixed-6x64": None, "stress-12x128": [3]}      # g18b: the BASELINE.json workloads (every aerodynamic phase but the last / one 128-node phase)


def case(name):
    """-> prob (with tau), D, x, {kind: spec rows (phase, range_all, limit)}, the G9 golden (or None)"""
    import oracle
    if name.startswith("g9_"):
        g = load_golden("g9_aero_example.npz")
        prob = dict(problem_from_golden(g))
        D, tau = D_tau_from_golden(g, prob)
        prob["tau"] = tau
        return prob, D, g["x"], {k: spec_from_golden(g, name[3:], k) for k in KINDS}, g
    if name in BASELINE:
        from gelato_amd import con_dynamics, pack_x, problem
        pdict, unitdict, condition, xdict = problem.make_problem(name)
        prob, x = dict(con_dynamics.problem_arrays(pdict, unitdict)), pack_x(xdict)
        P = oracle.Problem(prob)
        prob["tau"] = [P.tau(i) for i in range(P.S)]
        specs = {k: np.array([(i, 1, LIMITS[k]) for i in range(P.S - 1) if prob["reference_area"][i] != 0.0 and (BASELINE[name] is None or i in BASELINE[name])])
                 for k in KINDS}
        return prob, [P.D(i) for i in range(P.S)], x, specs, None
    build = {"ragged": states.ragged_state, "polar": lambda: states.with_coast_tail(states.polar_dense_state),
             "layers": lambda: states.with_coast_tail(states.all_layers_state),
             "breaks": lambda: states.with_coast_tail(states.layer_break_state)}[name]
    prob, x = build()
    P = oracle.Problem(prob)
    prob = dict(prob)
    prob["tau"] = [P.tau(i) for i in range(P.S)]
    specs = {k: np.array([(i, 1, LIMITS[k]) for i in range(P.S - 1) if prob["reference_area"][i] != 0.0]) for k in KINDS}
    return prob, [P.D(i) for i in range(P.S)], x, specs, None


class Truth:
    """the fixture's rows for one kind's spec, in the constraint's row order, and everything derived from them"""

    def __init__(self, name, prob, x, kind, spec):
        import oracle
        G = load_golden("g18b_aero_exact_fd_baseline.npz" if name in BASELINE else "g18_aero_exact_fd.npz")
        assert np.array_equal(x, G[name + "_x"]), "the state builder no longer reproduces the fixture's decision vector"
        nn = [int(v) for v in prob["num_nodes"]]
        where, i = {}, 0
        for ph, al in G[name + "_nodes"]:
            for k in range(nn[ph] + 1 if al else 1):
                where[(int(ph), k)] = i
                i += 1
        idx, lim, self.blocks = [], [], []
        for ph, al, limit in spec:
            nk = nn[int(ph)] + 1 if int(al) else 1
            self.blocks.append((len(idx), nk))
            idx += [where[(int(ph), k)] for k in range(nk)]
            lim += [limit] * nk
        idx, self.lim = np.array(idx), np.array(lim)
        a, q, da, dq = (G["%s_%s" % (name, k)][idx] for k in ("alpha", "q", "d_alpha", "d_q"))
        dx = float(prob["dx"])
        # the quotient of q alpha from the quotients of q and alpha: exact (product rule of a finite difference)
        quot = {"alpha": da, "q": dq, "qalpha": q[:, None] * da + a[:, None] * dq + da * dq * dx}[kind] / self.lim[:, None]
        self.f = {"alpha": a, "q": q, "qalpha": q * a}[kind] / self.lim
        self.jac = -quot                                   # con = 1 - f (con_aero.py:127-139)
        self.kind, self.dx = kind, dx
        self.terms = fd_noise.aero_noise_terms(oracle, prob, x, spec)
        assert np.allclose(self.terms["alpha"], a, rtol=0, atol=1e-9) and np.allclose(self.terms["q"], q, rtol=1e-9, atol=1e-12)

    def value_bound(self):
        """per row: what one fp64 evaluation of f = alpha / limit, q / limit or q alpha / limit may be off the exact value (the e_f
        of tests/fd_noise.py: acos(c) at c -> 1 moves alpha by eps / sin(alpha) per ulp of c) -- beside the flat 1e-11 where the
        angle of attack is a few 1e-5 rad (the vertical ascent of the BASELINE meshes)"""
        b = fd_noise.aero_bound(self.terms, self.kind, self.lim, self.dx, position=False) * self.dx / 2.0
        b = np.where(np.isnan(b), 0.0, b)
        # rows without a finite bound (the air-relative speed may vanish within the wind table's range -- lift-off: alpha is noise
        # there) are COUNTED (self.unbounded_rows; the tests assert the count per fixture) and get the cap, not a free pass
        # (with inf left in, a regression in those rows' values could not fail)
        self.unbounded_rows = int(np.count_nonzero(~np.isfinite(b)))
        return np.minimum(b, VALUE_BOUND_CAP)

    def coo_order(self, per_row_cols):
        """[R, w] -> the block's values in the reference's emission order: per spec, component-major (con_aero.py:437-463)"""
        return np.concatenate([per_row_cols[r0:r0 + nk].T.ravel() for r0, nk in self.blocks]) if self.blocks else np.zeros(0)

    def check(self, var, vals, what, with_drift):
        if self.kind == "q" and var == "quaternion":
            assert vals.size == 0
            return 0.0
        exact = self.jac[:, COLS[var]]
        w = exact.shape[1]
        b = fd_noise.aero_bound(self.terms, self.kind, self.lim, self.dx, position=(var == "position"))
        if with_drift:
            b = b + fd_noise.aero_drift(self.terms, np.abs(self.jac[:, :10]), self.dx)
        bound = self.coo_order(np.repeat(b[:, None], w, axis=1)) + 1e-9 * np.abs(self.coo_order(exact))
        err = np.abs(vals - self.coo_order(exact))
        with np.errstate(invalid="ignore"):
            ratio = np.where(np.isfinite(bound), err / (bound + 1e-300), 0.0)
        assert ratio.max() <= 1.0, "%s %s/%s: |entry - exact| is %.2f x its bound (err %.3e, |exact| up to %.3g)" % (
            what, self.kind, var, ratio.max(), err[ratio.argmax()], np.abs(exact).max())
        self.last_flat_excess = err - (1e-5 + 1e-6 * np.abs(self.coo_order(exact)))     # for tests/parity_margin.py
        return ratio.max()


@pytest.mark.parametrize("name", CASES)
def test_oracle_and_reference_within_the_derived_bound_of_the_exact_quotients(name):
    import oracle
    prob, D, x, specs, g = case(name)
    P = oracle.Problem(prob, D=D, tau=prob["tau"])
    used = []
    for kind in KINDS:
        spec = specs[kind]
        if len(spec) == 0:
            continue
        P.aero_configure(kind, spec)
        T = Truth(name, prob, x, kind, spec)
        con = P.aero_residual(kind, x)
        assert np.all(np.abs(con - (1.0 - T.f)) <= 1e-11 + 1e-10 * np.abs(T.f) + T.value_bound()), (kind, np.abs(con - (1.0 - T.f)).max())
        assert T.unbounded_rows == UNBOUNDED_ROWS.get((name, kind), 0), (name, kind, T.unbounded_rows)
        J = P.aero_jacobian(kind, x)
        for var in VARS:
            used.append(T.check(var, J[var]["coo"][2], "oracle", with_drift=True))
            if g is not None:   # the values the imported reference produced (G9)
                used.append(T.check(var, g["%s_%s_jac_%s_vals" % (name[3:], kind, var)], "reference (G9)", with_drift=True))
    assert max(used) > 0.02, "the bound is vacuous here: nothing uses 2 % of it"


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 8])
@pytest.mark.parametrize("name", CASES)
def test_engine_within_the_derived_bound_of_the_exact_quotients(name, flags):
    from gelato_amd import Engine
    prob, D, x, specs, g = case(name)
    E = Engine(prob, D=D, tau=prob["tau"], flags=flags)
    for kind in KINDS:
        spec = specs[kind]
        E.aero_configure(kind, spec)
        if len(spec) == 0:
            continue
        T = Truth(name, prob, x, kind, spec)
        con, jv, rc = E.eval_aero(kind, x[None, :])
        assert rc == 0
        assert np.all(np.abs(con[0] - (1.0 - T.f)) <= 1e-11 + 1e-10 * np.abs(T.f) + T.value_bound())
        assert T.unbounded_rows == UNBOUNDED_ROWS.get((name, kind), 0), (name, kind, T.unbounded_rows)
        nrow, nnz = E.aero_dims(kind)
        off = 0
        for v, var in enumerate(VARS):
            vals = jv[0, off:off + nnz[v]]
            off += nnz[v]
            T.check(var, vals, "engine (flags %d)" % flags, with_drift=False)
            if var == "t" and flags == 0:
                assert not vals.any(), "the default form writes the exact t0 / tf columns: zeros"
