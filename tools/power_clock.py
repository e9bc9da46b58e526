#!/usr/bin/env python3
"""Shader clock and package power of the chip under the fused launch (GPU box): a child process loops the launch while this one
samples `rocm-smi --showclocks --showpower` -- the evidence for what bounds the kernel (the 1400 W package power
cap).  python3 tools/power_clock.py [workload ...] > profiles/r04/power_clock.json"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sample():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    g = lambda pat: (lambda m: float(m.group(1)) if m else None)(re.search(pat, o))   # noqa: E731
    return {"sclk_mhz": g(r"sclk clock level: \d+: \((\d+)Mhz\)"), "mclk_mhz": g(r"mclk clock level: \d+: \((\d+)Mhz\)"),
            "fclk_mhz": g(r"fclk clock level: \d+: \((\d+)Mhz\)"), "package_power_w": g(r"Power \(W\): ([0-9.]+)")}


if __name__ == "__main__":
    cap = subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout
    m = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", cap)
    out = {"power_cap_w": float(m.group(1)) if m else None, "idle": sample(), "workloads": []}
    variants = [("fused launch", []), ("residual only", ["--residual-only"])]
    for wl in sys.argv[1:] or ["mixed-6x64", "dense-6x64", "stress-12x128"]:
        for vname, extra in variants:
            B = "16384" if wl.startswith("stress") else "65536"
            code = ("import sys, time; sys.path.insert(0, %r); import numpy as np, torch\n"
                    "from gelato_amd import Engine, con_dynamics, pack_x, problem\n"
                    "pd, ud, c, xd = problem.make_problem(%r); E = Engine(con_dynamics.problem_arrays(pd, ud)); B = %s\n"
                    "X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64, 1)); dX = torch.from_numpy(X).cuda()\n"
                    "r = torch.empty((B, E.nres), dtype=torch.float64, device='cuda'); j = torch.empty((B, E.V), dtype=torch.float64, device='cuda')\n"
                    "s = torch.cuda.current_stream().cuda_stream; jp = 0 if %r else j.data_ptr()\n"
                    "print('READY', flush=True); t0 = time.time(); n = 0\n"
                    "while time.time() - t0 < 9.0:\n"
                    "    for _ in range(20): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), jp, s)\n"
                    "    torch.cuda.synchronize(); n += 20\n"
                    "print('EVALS_PER_S', n * B / (time.time() - t0), flush=True)\n") % (ROOT, wl, B, bool(extra))
            pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
            assert pr.stdout.readline().strip() == "READY"
            time.sleep(3.0)
            samples = []
            for _ in range(5):
                samples.append(sample())
                time.sleep(0.8)
            rate = None
            for line in pr.stdout:
                if line.startswith("EVALS_PER_S"):
                    rate = float(line.split()[1])
            pr.wait()
            out["workloads"].append({"workload": wl, "launch": vname, "batch": int(B), "evals_per_s_over_the_loop": rate, "samples_under_load": samples})
            time.sleep(2.0)
    print(json.dumps(out, indent=1))
