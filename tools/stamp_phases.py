#!/usr/bin/env python3
"""Where a wavefront of the fused launch spends its lifetime (diagnostic build -DGEL_STAMP: s_memtime stamps at eight points of
the kernel, lane 0 of every wavefront).  GPU box:  GELATO_AMD_LIB=build/variants/libgel_stamp.so python3 tools/stamp_phases.py [workload] [B]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
from gelato_amd._lib import lib
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
pdict, unitdict, condition, xdict = problem.make_problem(wl)
prob = con_dynamics.problem_arrays(pdict, unitdict)
E = Engine(prob)
dev = torch.device("cuda:0")
X = problem.synthetic_batch(pack_x(xdict), E.M, 256)
X = np.tile(X, (B // 256 + 1, 1))[:B]
dX = torch.from_numpy(X).to(dev)
dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
resonly = os.environ.get("RES_ONLY", "0") == "1"
for _ in range(100):
    E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), 0 if resonly else djv.data_ptr(), s)
torch.cuda.synchronize()
n = (1 << 18) * 8
buf = np.zeros(n, dtype=np.uint64)
L = lib()
L.gel_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.gel_debug_stamps(buf.ctypes.data, n) == 0
st = buf.reshape(-1, 8).astype(np.int64)
nw = min(1 << 18, E.launch_info(B, True, not resonly)[3])
st = st[:nw]
ok = (st[:, 7] > st[:, 0]) & (st[:, 0] > 0)
st = st[ok]
names = ["entry -> descriptors", "-> operands staged (barrier 1)", "-> D.X product", "-> hand-over (barriers 2, 3)",
         "-> mass / position / quaternion groups written", "-> centre + light sweeps", "-> position sweeps, end"]
d = np.diff(st, axis=1)
life = st[:, 7] - st[:, 0]
print("%s B=%d: %d wavefronts stamped, lifetime median %.0f cycles (p10 %.0f, p90 %.0f)" % (wl, B, len(st), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for i, nm in enumerate(names):
    col = d[:, i]
    col = col[(col >= 0) & (col < 10_000_000)]
    print("  %-50s median %8.0f  p10 %8.0f  p90 %8.0f   (%4.1f %% of the lifetime)" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90), 100 * np.median(col) / np.median(life)))
