#!/usr/bin/env python3
"""Where a wavefront of the fused launch spends its lifetime, and what the OTHER wavefronts of its SIMD do meanwhile (diagnostic build
-DGEL_STAMP, tools/ablations/stamp_build.patch: s_memtime stamps at eight points of the kernel + the hardware id of every wavefront).
GPU box:  GELATO_AMD_LIB=build/variants/libgel_stamp.so python3 tools/stamp_phases.py [workload] [B]      env AERO=1: the AERO instantiation
(gel_eval_batch_aero_device), RES_ONLY=1: residual-only launches"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelato_amd import Engine, con_dynamics, pack_x, problem
from gelato_amd._lib import lib
wl = sys.argv[1] if len(sys.argv) > 1 else "mixed-6x64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
aero = os.environ.get("AERO", "0") == "1"
resonly = os.environ.get("RES_ONLY", "0") == "1"
pdict, unitdict, condition, xdict = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pdict, unitdict))
dev = torch.device("cuda:0")
X = np.tile(problem.synthetic_batch(pack_x(xdict), E.M, 256), (B // 256 + 1, 1))[:B]
dX = torch.from_numpy(X).to(dev)
dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
djv = torch.empty((B, E.V), dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
if aero:
    S = pdict["num_sections"]
    for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
        E.aero_configure(kind, [(i, 1, lim) for i in range(S - 1)])
    da = torch.empty((B, E.aero_record_layout()[0]), dtype=torch.float64, device=dev)
for _ in range(60):
    if aero:
        E.eval_batch_aero_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), da.data_ptr(), s)
    else:
        E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), 0 if resonly else djv.data_ptr(), s)
torch.cuda.synchronize()
n = 1 << 18
buf = np.zeros(n * 8, dtype=np.uint64)
hw = np.zeros(n, dtype=np.uint32)
L = lib()
f_t, f_h = (L.gel_debug_stamps_aero, L.gel_debug_stamp_hw_aero) if aero else (L.gel_debug_stamps, L.gel_debug_stamp_hw)
f_t.argtypes = [C.c_void_p, C.c_size_t]; f_h.argtypes = [C.c_void_p, C.c_size_t]
assert f_t(buf.ctypes.data, n * 8) == 0 and f_h(hw.ctypes.data, n) == 0
st = buf.reshape(-1, 8).astype(np.int64)
nw = min(n, E.launch_info(B, True, not resonly)[3])
st, hw = st[:nw], hw[:nw]
ok = (st[:, 7] > st[:, 0]) & (st[:, 0] > 0) & np.all(np.diff(st, axis=1) >= 0, axis=1)
st, hw = st[ok], hw[ok]
names = ["entry -> descriptors", "-> operands staged (barrier 1)", "-> D.X product", "-> hand-over (barriers 2, 3)",
         "-> mass / position / quaternion groups written", "-> centre + light sweeps" + (" + aero rows of the centre" if aero else ""),
         "-> position sweeps" + (" + their aero rows" if aero else "") + ", end"]
d = np.diff(st, axis=1)
life = st[:, 7] - st[:, 0]
print("%s B=%d%s: %d wavefronts stamped (last launch), lifetime median %.0f cycles (p10 %.0f, p90 %.0f)"
      % (wl, B, " AERO" if aero else (" residual-only" if resonly else ""), len(st), np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
for i, nm in enumerate(names):
    col = d[:, i]
    print("  %-62s median %8.0f  p10 %8.0f  p90 %8.0f   (%4.1f %% of the lifetime)" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90), 100 * np.median(col) / np.median(life)))
# ---- what shares a SIMD: per SIMD (XCC, SE, SH, CU, SIMD of the hardware id) the stage every resident wavefront is in, sampled in time
simd_key = (hw >> 4) & 0xfff | ((hw >> 16) << 12)       # HW_ID: [3:0] wave slot, [5:4] SIMD, [7:6] pipe, [11:8] CU, [12] SH, [15:13] SE; XCC above
classes = ["waiting for operands (entry .. staged)", "D.X product + hand-over", "groups written", "centre + light sweeps", "position sweeps"]
edges = [0, 2, 4, 5, 6, 7]
rng = np.random.default_rng(0)
keys = np.unique(simd_key)
pick = rng.choice(keys, size=min(len(keys), 256), replace=False)
res_count = np.zeros(9); cls_time = np.zeros(len(classes)); idle_like = 0.0; total = 0.0; same = 0.0; joint = np.zeros((5, 5))
for k in pick:
    w = st[simd_key == k]
    if len(w) < 8:
        continue
    lo, hi = np.percentile(w[:, 0], 10), np.percentile(w[:, 7], 90)      # away from the launch's ramp and tail
    ts = np.linspace(lo, hi, 400)
    for t in ts:
        live = w[(w[:, 0] <= t) & (w[:, 7] > t)]
        if len(live) == 0:
            continue
        c = np.array([np.searchsorted(row[edges], t, side="right") - 1 for row in live])
        c = np.clip(c, 0, 4)
        res_count[min(len(live), 8)] += 1
        total += 1
        for ci in c:
            cls_time[ci] += 1.0 / len(live)
        idle_like += float(np.all(c == 0))
        same += float(len(live) > 1 and np.all(c == c[0]))
        for a_ in c:
            for b_ in c:
                joint[a_, b_] += 1.0 / (len(live) ** 2)
print("per SIMD (%d SIMDs sampled, 400 instants each between the launch's ramp and tail):" % len(pick))
print("  resident wavefronts: " + ", ".join("%d: %.1f %%" % (i, 100 * res_count[i] / total) for i in range(1, 9) if res_count[i] > 0))
print("  share of a resident wavefront's time per stage: " + "; ".join("%s %.1f %%" % (classes[i], 100 * cls_time[i] / total) for i in range(5)))
print("  instants at which EVERY resident wavefront waits for its operands: %.1f %% (independent stages would give %.1f %%)"
      % (100 * idle_like / total, 100 * (cls_time[0] / total) ** 4))
print("  instants at which all resident wavefronts are in the SAME stage: %.1f %% (independent: %.1f %%)"
      % (100 * same / total, 100 * float(np.sum((cls_time / total) ** 4))))
