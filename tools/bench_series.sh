#!/bin/bash
# the same bench command several times in a row on one (fresh) box, then one long-lived process timing the launch every half second:
# does the figure depend on how long the box / the process has been busy?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys; d = json.loads(sys.stdin.read()); print('run $i: %.2f M evals/s  settled %.2f M  kernel_ms %.3f' % (d['value'] / 1e6, d['value_settled'] / 1e6, d['roofline']['kernel_ms']))"
done
python3 - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
pd, ud, c, xd = problem.make_problem("mixed-6x64"); E = Engine(con_dynamics.problem_arrays(pd, ud)); B = 65536
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64, 1)); dX = torch.from_numpy(X).cuda()
r = torch.empty((B, E.nres), dtype=torch.float64, device="cuda"); j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
t0 = time.time(); out = []
while time.time() - t0 < 25.0:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), j.data_ptr(), s)
    b.record(); torch.cuda.synchronize()
    out.append((time.time() - t0, a.elapsed_time(b) / 20))
print("one process, continuous launches: t [s] -> ms/launch")
for k in range(0, len(out), max(1, len(out) // 40)): print("  %5.1f  %.4f" % out[k])
PY
