#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 tests/parity_margin.py > gpurun_out/parity_margins.json 2> gpurun_out/parity_margins.err; tail -5 gpurun_out/parity_margins.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/parity_margins.json"))
print(json.dumps(d["summary"], indent=1))
for r in d["aero"]:
    if r["entries_needing_derived_allowance"] or (r["worst_flat_excess_benign"] or -1) > 0:
        print(r["fixture"], r["flags"], r["kind"], r["var"], r["against"], "need", r["entries_needing_derived_allowance"], "of", r["entries"], "maxd %.2e" % r["max_abs_diff"], "benign", r["worst_flat_excess_benign"], r["worst_flat_excess_by_alpha_deg"], "minalpha %.2f" % r["min_alpha_deg"])
for r in d["defect"]:
    if r["entries_needing_derived_allowance"]:
        print(r["fixture"], r["flags"], r["block"], r["against"], "need", r["entries_needing_derived_allowance"], "of", r["entries"], "maxd %.2e" % r["max_abs_diff"], "allow %.2e" % r["derived_allowance_max"], "benign", r.get("worst_flat_excess_benign"))
PY
timeout 1700 python -m pytest tests -m gpu -x -q --deselect tests/test_aero_engine.py::test_aero_values_and_gradients_gpu 2>&1 | tail -15
