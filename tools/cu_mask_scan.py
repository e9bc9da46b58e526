#!/usr/bin/env python3
"""The fused launch on a SUBSET of the compute units (HSA_CU_MASK in the child's environment; hipExtStreamCreateWithCUMask and
ROC_GLOBAL_CU_MASK are accepted on this pool but change nothing), with the package power and the shader clock
sampled beside it (GPU box).  If the full chip is held back by its package power cap, the same kernel on half of the CUs draws less,
clocks higher and does MORE per CU; if it is held back by the memory system, per-CU throughput rises because the memory system is
shared by fewer; if by its own dependency chains, per-CU throughput stays.  python3 tools/cu_mask_scan.py [workload] > cu_mask_scan.json"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.power_clock import sample   # noqa: E402

MASKS = [("all 256", None), ("CUs 0-191", "0:0-191"), ("CUs 0-127", "0:0-127"), ("CUs 0-63", "0:0-63")]
CHILD = r"""
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
wl, B = sys.argv[1], int(sys.argv[2])
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64, 1)); dX = torch.from_numpy(X).cuda()
r = torch.empty((B, E.nres), dtype=torch.float64, device='cuda'); j = torch.empty((B, E.V), dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
for _ in range(5): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
torch.cuda.synchronize()
print('READY', flush=True); t0 = time.time(); n = 0
while time.time() - t0 < 8.0:
    for _ in range(10): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
    torch.cuda.synchronize(); n += 10
print('EVALS_PER_S', n * B / (time.time() - t0), flush=True)
""" % ROOT

if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
    B = 16384 if wl.startswith("stress") else 65536
    out = {"workload": wl, "batch": B, "idle": sample(), "rows": []}
    for name, mask in MASKS:
        env = dict(os.environ) if mask is None else dict(os.environ, HSA_CU_MASK=mask)
        pr = subprocess.Popen([sys.executable, "-c", CHILD, wl, str(B)], env=env, stdout=subprocess.PIPE, text=True)
        line = pr.stdout.readline().strip()
        assert line == "READY", line
        time.sleep(2.5)
        samples = [sample() for _ in range(4)]
        rate = None
        for line in pr.stdout:
            if line.startswith("EVALS_PER_S"):
                rate = float(line.split()[1])
        pr.wait()
        cus = 256 if mask is None else int(mask.split("-")[-1]) + 1
        out["rows"].append({"mask": name, "cus": cus, "evals_per_s": rate, "evals_per_s_per_cu": rate / cus if rate else None,
                            "sclk_mhz": sum(s["sclk_mhz"] or 0 for s in samples) / len(samples),
                            "package_power_w": sum(s["package_power_w"] or 0 for s in samples) / len(samples)})
        time.sleep(2.0)
    full = out["rows"][0]["evals_per_s_per_cu"]
    for r in out["rows"]:
        r["per_cu_vs_full_chip"] = r["evals_per_s_per_cu"] / full if r["evals_per_s_per_cu"] and full else None
    print(json.dumps(out, indent=1))
