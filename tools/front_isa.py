#!/usr/bin/env python3
"""(here, no GPU) memory instructions, waits, barriers and MFMAs of one eval_kernel instantiation in program order, runs of the same
opcode compressed: where the first waits of a wavefront sit.  tools/front_isa.py [mangled-prefix] [max rows] [listing.s]"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gelato_amd", "csrc")
prefix = sys.argv[1] if len(sys.argv) > 1 else "_ZN3gel11eval_kernelILb1ELb1ELb0ELb0EE"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lst = sys.argv[3] if len(sys.argv) > 3 else "/tmp/gel_front.s"
if len(sys.argv) <= 3:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-mllvm",
                    "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--cuda-device-only", "-S", "gel_kernels.hip", "-o", lst], cwd=root, check=True, stderr=subprocess.DEVNULL)
L = open(lst).read().split("\n")
start = [i for i, l in enumerate(L) if l.startswith(prefix) and ": ;" in l][0]
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
pat = re.compile(r"(global_load|buffer_load|s_load|s_waitcnt|s_barrier|v_mfma|ds_write|ds_read|v_readfirstlane|global_store|buffer_store|s_cbranch|s_branch|\.LBB)")
res, prev, cnt, first = [], None, 0, 0
for i in range(start, end):
    l = L[i].strip()
    if not pat.match(l):
        continue
    l = l.split(";")[0].strip()
    op = l.split()[0]
    key = l if (op.startswith("s_waitcnt") or op.startswith(".LBB") or "branch" in op) else op
    if key == prev:
        cnt += 1
    else:
        if prev is not None:
            res.append((first, prev, cnt))
        prev, cnt, first = key, 1, i - start
res.append((first, prev, cnt))
for r in res[:rows]:
    print("%6d  %-48s x%d" % r)
