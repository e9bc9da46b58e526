#!/usr/bin/env python3
"""How much of every tolerance the parity suite really uses (GPU box):   python tests/parity_margin.py > profiles/r04/parity_margins.json

Per fixture and per COO block: max |GPU - reference|, the FLAT allowance of SURVEY 8(c) (1e-5 + 1e-6 |ref|) and how many
entries exceed it, the DERIVED allowance where a test uses one (tests/fd_noise.py: the reference's own finite-difference noise at
that node) and the worst excess over it, and the worst flat excess in the benign region (|latitude| < 55 deg; aero rows: and an
angle of attack above 1 deg) where the tests insist on the flat tolerance alone.  The same functions the tests assert on
(test_gpu_parity.defect_margins, test_aero_engine.aero_margins), so the table cannot drift from what is tested."""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from conftest import D_tau_from_golden, load_golden, problem_from_golden
import states
import test_aero_engine as ta
import test_gpu_parity as tg
from gelato_amd import _lib

out = {"build": _lib.build_info(), "flat_tolerance": "1e-5 + 1e-6 |ref|", "benign_region": "|lat| < %g deg (aero rows: and alpha > %g deg)" % (tg.BENIGN_LAT_DEG, ta.BENIGN_ALPHA_DEG),
       "defect": [], "aero": []}
oracle = tg._setup()


def add(fixture, flags, against, table):
    for row in table:
        out["defect"].append(dict({"fixture": fixture, "flags": flags, "against": against}, **row))


# G6: the reference's own outputs (golden) -- and the oracle on the same problem -- default form and GEL_FLAG_FD_RECOMPUTE
for name in ["example", "3x32", "mixed6x64", "dense6x64", "negarea"]:
    g = load_golden("g6_%s.npz" % name)
    prob = problem_from_golden(g)
    D, tau = D_tau_from_golden(g, prob)
    for flags in (0, 8):
        E, P = tg.make_pair(prob, D, tau, barC20=tg.TW, flags=flags)
        table, res, vals = tg.defect_margins(E, P, g["x"])
        add("g6_" + name, flags, "oracle", table)
        J = E.jac_dicts(vals)
        for grp in oracle.GROUPS:
            for var in oracle.BLOCK_VARS[grp]:
                key = "jac_%s_%s" % (grp, var)
                v = J[grp][var]["coo"][2]
                if key + "_vals" in g:
                    ref, got = g[key + "_vals"], v
                elif "var_%s_%s_idx" % (grp, var) in g:
                    ref, got = g["var_%s_%s_vals" % (grp, var)], v[g["var_%s_%s_idx" % (grp, var)]]
                else:
                    continue
                d = np.abs(got - ref); flat = 1e-5 + 1e-6 * np.abs(ref)
                out["defect"].append({"fixture": "g6_" + name, "flags": flags, "against": "reference (G6 golden)", "block": "%s/%s" % (grp, var),
                                      "entries": int(d.size), "max_abs_diff": float(d.max()), "max_abs_ref": float(np.abs(ref).max()),
                                      "worst_flat_excess": float((d - flat).max()), "entries_needing_derived_allowance": int(np.count_nonzero(d > flat)),
                                      "derived_allowance_max": 0.0})
        E.close()
# BASELINE workloads at full size, against the oracle, both forms
for name in ["dense-6x64", "mixed-6x64", "stress-12x128"]:
    prob, x, _ = tg.named_problem(name)
    for flags in (0, 8):
        E, P = tg.make_pair(prob, flags=flags)
        add(name, flags, "oracle", tg.defect_margins(E, P, x)[0])
        E.close()
# the extreme states: where the derived allowance is in use
for name, build in [("ragged", states.ragged_state), ("polar-dense", states.polar_dense_state), ("all-layers", states.all_layers_state),
                    ("layer-breaks", states.layer_break_state)]:
    prob, x = build()
    for flags in (0, 8):
        E, P = tg.make_pair(prob, flags=flags)
        add(name, flags, "oracle", tg.defect_margins(E, P, x, prob)[0])
        E.close()
for cname in ("example", "synthetic"):
    for flags in (0, 8):
        out["aero"] += ta.aero_margins(cname, flags)
# how far the REFERENCE's own aero gradients (G9) and the oracle are from the exact quotients (G18), against the flat tolerance: the
# example flies at 0.1 .. 2.5 degrees of angle of attack, where alpha = acos(c) carries eps / sin(alpha) / dx per evaluation --
# no fp64 implementation of that formula meets 1e-5 + 1e-6 |ref| there, the reference included
import oracle as _orc
import test_aero_exact_fd as tx
out["aero_vs_exact"] = []
for name in ("g9_example", "g9_synthetic"):
    prob, D, x, specs, g = tx.case(name)
    Po = _orc.Problem(prob, D=D, tau=prob["tau"])
    for kind in tx.KINDS:
        spec = specs[kind]
        if len(spec) == 0:
            continue
        Po.aero_configure(kind, spec)
        T = tx.Truth(name, prob, x, kind, spec)
        Jo = Po.aero_jacobian(kind, x)
        for var in tx.VARS:
            for who, vals in (("reference (G9)", g["%s_%s_jac_%s_vals" % (name[3:], kind, var)]), ("oracle", Jo[var]["coo"][2])):
                if vals.size == 0:
                    continue
                T.check(var, vals, who, with_drift=True)
                out["aero_vs_exact"].append({"fixture": name, "kind": kind, "var": var, "who": who, "entries": int(vals.size),
                                             "entries_beyond_flat_tolerance_of_the_exact_quotient": int(np.count_nonzero(T.last_flat_excess > 0)),
                                             "worst_flat_excess": float(T.last_flat_excess.max())})
# summary
dd = out["defect"]
out["summary"] = {
    "defect_blocks": len(dd), "defect_entries_needing_derived_allowance": int(sum(r["entries_needing_derived_allowance"] for r in dd)),
    "defect_worst_flat_excess_where_no_allowance_is_used": max(r["worst_flat_excess"] for r in dd if r.get("derived_allowance_max", 0.0) == 0.0),
    "aero_blocks": len(out["aero"]), "aero_entries_needing_derived_allowance": int(sum(r["entries_needing_derived_allowance"] for r in out["aero"])),
    "aero_worst_flat_excess_benign": max((r["worst_flat_excess_benign"] for r in out["aero"] if r["worst_flat_excess_benign"] is not None), default=None),
    "reference_G9_entries_beyond_flat_tolerance_of_the_exact_quotient": int(sum(r["entries_beyond_flat_tolerance_of_the_exact_quotient"]
                                                                                 for r in out["aero_vs_exact"] if r["who"].startswith("reference"))),
    "reference_G9_entries": int(sum(r["entries"] for r in out["aero_vs_exact"] if r["who"].startswith("reference")))}
print(json.dumps(out, indent=1))
