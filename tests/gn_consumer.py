"""A consumer of the callback surface (test infrastructure): damped Gauss-Newton feasibility steps on the shipped example,
driven through ``objfunc`` / ``sens`` exactly the way pyoptsparse drives them (Trajectory_Optimization.py:354-355,454-458) --
every group's values and every Jacobian block of the 23 keys are placed into ONE scipy.sparse matrix over the packed
decision vector and a linear system is solved with it.  It shows that the assembled Jacobian is right AS A MATRIX: a step
computed from it reduces the constraint violation, and the iterates follow the same loop driven by the oracle.

  residual rows   every eqcon_* group, and the violated rows of every ineqcon_* group (g < 0)
  step            min || A dx + r ||^2 + mu || dx ||^2  by sparse Cholesky-free normal equations (SuperLU), mu fixed
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

XKEYS = ["mass", "position", "velocity", "quaternion", "u", "t"]


def _offsets(xdict):
    sizes = [np.asarray(xdict[k]).size for k in XKEYS]
    return dict(zip(XKEYS, np.concatenate([[0], np.cumsum(sizes)[:-1]]))), int(sum(sizes))


def assemble(funcs, fs, xdict):
    """-> (r [m], A [m x nvars] CSR, names of the groups used, objective gradient [nvars])"""
    off, nv = _offsets(xdict)
    blocks, rs, used = [], [], []
    for key in sorted(funcs):
        if key == "obj" or funcs[key] is None:
            continue
        val = np.atleast_1d(np.asarray(funcs[key], dtype=np.float64))
        J = sp.lil_matrix((len(val), nv)) if False else None
        rows_all, cols_all, vals_all = [], [], []
        for var, blk in fs[key].items():
            if isinstance(blk, dict):                                   # {"coo": [rows, cols, vals], "shape": ...}
                r_, c_, v_ = blk["coo"]
                assert blk["shape"] == (len(val), np.asarray(xdict[var]).size), (key, var, blk["shape"])
                rows_all.append(np.asarray(r_)); cols_all.append(np.asarray(c_) + off[var]); vals_all.append(np.asarray(v_))
            else:                                                       # dense [rows, size] (lib/jac_fd.py)
                d = np.asarray(blk)
                assert d.shape == (len(val), np.asarray(xdict[var]).size), (key, var, d.shape)
                rr, cc = np.nonzero(d)
                rows_all.append(rr); cols_all.append(cc + off[var]); vals_all.append(d[rr, cc])
        A = sp.coo_matrix((np.concatenate(vals_all), (np.concatenate(rows_all), np.concatenate(cols_all))), shape=(len(val), nv)).tocsr()
        if key.startswith("ineqcon"):
            act = val < 0.0                                             # only the violated rows pull
            if not act.any():
                used.append(key + " (satisfied)")
                continue
            A, val = A[np.nonzero(act)[0]], val[act]
        blocks.append(A); rs.append(val); used.append(key)
    g = np.zeros(nv)
    for var, blk in fs["obj"].items():
        g[off[var]:off[var] + np.asarray(blk).size] = np.asarray(blk).ravel()
    return np.concatenate(rs), sp.vstack(blocks).tocsr(), used, g


def gauss_newton(objfunc, sens, xdict0, iterations=6, mu=1e-6):
    """-> list of per-iteration records {x (packed), norm (|| r ||), groups}; the last record is the final point"""
    x = {k: np.array(xdict0[k], dtype=np.float64) for k in XKEYS}
    off, nv = _offsets(x)
    trace = []
    for it in range(iterations + 1):
        funcs, fail = objfunc(x)
        assert not fail
        fs, fail = sens(x, funcs)
        assert not fail
        r, A, used, g = assemble(funcs, fs, x)
        trace.append({"x": np.concatenate([x[k].ravel() for k in XKEYS]), "norm": float(np.linalg.norm(r)), "groups": used,
                      "rows": int(A.shape[0]), "nnz": int(A.nnz), "obj": float(funcs["obj"]), "gdotdx": None})
        if it == iterations:
            break
        H = (A.T @ A + mu * sp.identity(nv)).tocsc()
        dx = spla.splu(H).solve(-(A.T @ r))
        trace[-1]["gdotdx"] = float(g @ dx)
        for k in XKEYS:
            x[k] = x[k] + dx[off[k]:off[k] + x[k].size].reshape(x[k].shape)
    return trace


def oracle_callbacks(pdict, unitdict, condition):
    """objfunc / sens with the reference's 23 keys, every group from the CPU oracle (oracle/gelato_oracle.c, knot_terminal.py,
    waypoint.py): the independent driver of the same loop."""
    import oracle
    from oracle import knot_terminal as kt
    from oracle import waypoint as wp
    from gelato_amd import con_aero, con_dynamics, pack_x     # problem flattening and the aero spec reader: data plumbing, no arithmetic
    from gelato_amd.cost_gradient import cost_6DoF, cost_jac
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    ps = pdict["ps_params"]
    S = pdict["num_sections"]
    P = oracle.Problem(prob, D=[ps.D(i) for i in range(S)], tau=[ps.tau(i) for i in range(S)])
    spc = kt.make_spec(pdict, unitdict, condition)
    wrows = wp.make_rows(spc, pdict, condition)
    kinds = {"alpha": "ineqcon_alpha", "q": "ineqcon_q", "qalpha": "ineqcon_qalpha"}
    nspec = {}
    for kind in kinds:
        spec = con_aero._spec(pdict, condition, kind)
        nspec[kind] = len(spec)
        if len(spec):
            P.aero_configure(kind, spec)
    user_section = pdict["event_index"]["IIP_END"]
    groups = {"mass": "eqcon_dyn_mass", "pos": "eqcon_dyn_pos", "vel": "eqcon_dyn_vel", "quat": "eqcon_dyn_quat"}
    wgroups = {"eqpos": "eqcon_pos", "ineqpos": "ineqcon_pos", "eqiip": "eqcon_iip", "ineqiip": "ineqcon_iip", "antenna": "ineqcon_antenna"}

    def objfunc(xdict):
        x = pack_x(xdict)
        f = {"obj": cost_6DoF(xdict, condition)}
        f["eqcon_init"] = kt.equality_init(x, spc); f["eqcon_time"] = kt.equality_time(x, spc)
        f["eqcon_knot"] = kt.equality_knot_LGR(x, spc); f["eqcon_terminal"] = kt.equality_terminal(x, spc)
        f["eqcon_rate"] = kt.equality_rate(x, spc); f["ineqcon_time"] = kt.inequality_time(x, spc)
        f["ineqcon_mass"] = kt.inequality_mass(x, spc); f["ineqcon_kick"] = kt.inequality_kickturn(x, spc)
        f["eqcon_user"] = kt.user_apogee_height(x, spc, user_section); f["ineqcon_user"] = None
        for g, key in groups.items():
            f[key] = P.residual(g, x)
        for g, key in wgroups.items():
            f[key] = wp.values(x, spc, wrows, g)
        for kind, key in kinds.items():
            f[key] = P.aero_residual(kind, x) if nspec[kind] else None
        return f, False

    def sens(xdict, funcs):
        x = pack_x(xdict)
        fs = {"obj": cost_jac(xdict, condition)}
        fs["eqcon_init"] = kt.equality_jac_init(x, spc); fs["eqcon_time"] = kt.equality_jac_time(x, spc)
        fs["eqcon_knot"] = kt.equality_jac_knot_LGR(x, spc); fs["eqcon_terminal"] = kt.equality_jac_terminal(x, spc)
        fs["eqcon_rate"] = kt.equality_jac_rate(x, spc); fs["ineqcon_time"] = kt.inequality_jac_time(x, spc)
        fs["ineqcon_mass"] = kt.inequality_jac_mass(x, spc); fs["ineqcon_kick"] = kt.inequality_jac_kickturn(x, spc)
        Ju = kt.jac_fd_dense(lambda xx: kt.user_apogee_height(xx, spc, user_section), x, spc["dx"])
        o = np.cumsum([0] + [np.asarray(xdict[k]).size for k in XKEYS])
        fs["eqcon_user"] = {k: Ju[:, o[i]:o[i + 1]] for i, k in enumerate(XKEYS)}
        fs["ineqcon_user"] = None
        for g, key in groups.items():
            fs[key] = P.jacobian(g, x)
        for g, key in wgroups.items():
            J = wp.jacobian(x, spc, wrows, g)
            fs[key] = None if J is None else {var: {"coo": [r_, c_, v_], "shape": shp} for var, (r_, c_, v_, shp) in J.items()}
        for kind, key in kinds.items():
            fs[key] = P.aero_jacobian(kind, x) if nspec[kind] else None
        return fs, False

    return objfunc, sens
