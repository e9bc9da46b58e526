"""(GPU box) defect groups + aero rows: the one call (fused launch + the small launch for state node 0) against the two kernels, per
kernel.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split, or alone for HIP-event times.
usage: aero_fused_probe.py [workload] [B] [calls]      env GEL_AERO_UNFUSED=1: the one call falls back to the two kernels"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gelato_amd import Engine, con_dynamics, pack_x, problem
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
S = pd["num_sections"]
for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
    E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 256), (B // 256 + 1, 1))[:B]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
dX = torch.from_numpy(X).to(dev)
width, oc, oj = E.aero_record_layout()
r = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
j = torch.empty((B, E.V), dtype=torch.float64, device=dev)
a = torch.empty((B, width), dtype=torch.float64, device=dev)
dims = [E.aero_dims(k) for k in E.AERO_KINDS]
dcon = [torch.empty((B, d[0]), dtype=torch.float64, device=dev) for d in dims]
djac = [torch.empty((B, sum(d[1])), dtype=torch.float64, device=dev) for d in dims]
cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]


def t_of(fn, n=N, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = {"workload": wl, "batch": B, "record_doubles": width}
out["defect_ms"] = t_of(lambda: E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s))
out["aero_kernel_ms"] = t_of(lambda: E.eval_aero_all_device(B, dX.data_ptr(), cp, jp, s))
out["one_call_ms"] = t_of(lambda: E.eval_batch_aero_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), a.data_ptr(), s))
out["defect_ms_again"] = t_of(lambda: E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s))
out["one_call_ms_again"] = t_of(lambda: E.eval_batch_aero_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), a.data_ptr(), s))
out["ns_per_vector"] = {"two_kernels": 1e6 * (out["defect_ms"] + out["aero_kernel_ms"]) / B, "one_call": 1e6 * min(out["one_call_ms"], out["one_call_ms_again"]) / B}
print(json.dumps(out))
