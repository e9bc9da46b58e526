#!/usr/bin/env python3
"""A/B of two engine builds INSIDE one process: both libraries are loaded (the package imported twice, under two names), both
handles launch on the SAME device buffers, turns of AB_TURN_MS (default 1000) alternate in shuffled order.  Where the driver places
a process's 14 GB of buffers moves a launch by up to 7 % (tools/ab_variants.py: processes of ONE build 3.06 .. 3.29 ms on one box);
here the placement is common to both builds.  AB_PROCS (default 3) fresh processes = that many placements; pooled at the end.
GPU box:  python3 tools/ab_inproc.py <workload> <batch> <rounds> <libA.so> <libB.so>   [env AB_RES_ONLY=1 | AB_AERO=1 | AB_FUSED=1]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import importlib.util, os, random, sys, json
sys.path.insert(0, %r)
root = %r
wl, B, rounds, libs, res_only = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:6], sys.argv[6] == "1"
import numpy as np, torch
pk = []
for i, lib in enumerate(libs):
    os.environ["GELATO_AMD_LIB"] = lib
    name = "gelato_amd_ab%%d" %% i
    spec = importlib.util.spec_from_file_location(name, os.path.join(root, "gelato_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(root, "gelato_amd")])
    m = importlib.util.module_from_spec(spec); sys.modules[name] = m; spec.loader.exec_module(m)
    importlib.import_module(name + ".problem"); importlib.import_module(name + ".con_dynamics")
    pk.append(m)
E = []
for i, m in enumerate(pk):
    # AB_ENVA / AB_ENVB = "K=V,K=V": environment switches read at handle creation, per side (the same library may be given twice)
    for kv in filter(None, os.environ.get("AB_ENV" + "AB"[i], "").split(",")):
        k, v = kv.split("=", 1)
        os.environ[k] = v
    pd, ud, c, xd = m.problem.make_problem(wl)
    E.append(m.Engine(m.con_dynamics.problem_arrays(pd, ud)))
    for kv in filter(None, os.environ.get("AB_ENV" + "AB"[i], "").split(",")):
        os.environ.pop(kv.split("=", 1)[0], None)
m = pk[0]
X = np.tile(m.problem.synthetic_batch(m.pack_x(xd), E[0].M, 64), (B // 64 + 1, 1))[:B]
dX = torch.from_numpy(X).cuda()
r = torch.empty((B, E[0].nres), dtype=torch.float64, device="cuda"); j = torch.empty((B, E[0].V), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
jp = 0 if res_only else j.data_ptr()
aero = os.environ.get("AB_AERO", "0") == "1"          # the aero path constraints' kernel instead of the fused one
if aero:
    S = pd["num_sections"]
    for e in E:
        for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
            e.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
    dims = [E[0].aero_dims(k) for k in E[0].AERO_KINDS]
    dcon = [torch.empty((B, d[0]), dtype=torch.float64, device="cuda") for d in dims]
    djac = [torch.empty((B, sum(d[1])), dtype=torch.float64, device="cuda") for d in dims]
    cp, jpp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]
fused = os.environ.get("AB_FUSED", "0") == "1"         # defect groups + aero rows in one call (gel_eval_batch_aero_device)
if fused:
    S = pd["num_sections"]
    for e in E:
        for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
            e.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
    da = torch.empty((B, max(e.aero_record_layout()[0] for e in E)), dtype=torch.float64, device="cuda")   # the builds' records may differ in width
def burst(e, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    if fused:
        for _ in range(n): e.eval_batch_aero_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), da.data_ptr(), s)
    elif aero:
        for _ in range(n): e.eval_aero_all_device(B, dX.data_ptr(), cp, jpp, s)
    else:
        for _ in range(n): e.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), jp, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for e in E: burst(e, 5)
turn_ms = float(os.environ.get("AB_TURN_MS", "1000"))
n = max(4, int(turn_ms / burst(E[0], 8)))
ts = [[], []]
random.seed(os.getpid())
for rd in range(rounds):
    order = [0, 1]; random.shuffle(order)
    for i in order:
        burst(E[i], max(2, n // 3))          # untimed lead-in of the turn
        ts[i].append(burst(E[i], n))
print(json.dumps([float(np.median(t)) for t in ts]))
""" % (ROOT, ROOT)
wl, B, rounds, la, lb = sys.argv[1], sys.argv[2], sys.argv[3], os.path.abspath(sys.argv[4]), os.path.abspath(sys.argv[5])
out = []
for p in range(int(os.environ.get("AB_PROCS", "3"))):
    q = subprocess.run([sys.executable, "-c", CHILD, wl, B, rounds, la, lb, os.environ.get("AB_RES_ONLY", "0")], capture_output=True, text=True)
    try:
        out.append(json.loads(q.stdout.strip().splitlines()[-1]))
    except Exception:
        print(q.stderr[-2000:], file=sys.stderr); raise
a = np.array(out)
print(json.dumps({"workload": wl, "batch": int(B), "rounds": int(rounds), "A": os.path.basename(la), "B": os.path.basename(lb), "res_only": os.environ.get("AB_RES_ONLY", "0") == "1",
                  "median_ms_per_process": {"A": [round(v, 4) for v in a[:, 0]], "B": [round(v, 4) for v in a[:, 1]]},
                  "ratio_B_over_A_per_process": [round(float(v), 4) for v in a[:, 1] / a[:, 0]], "ratio_B_over_A": round(float(np.mean(a[:, 1] / a[:, 0])), 4)}))
