#!/usr/bin/env python3
"""(here, no GPU) where one eval_kernel instantiation needs its registers: built with a 256-VGPR budget, the source lines whose
instructions touch registers at or above a threshold (default v128).  tools/vgpr_map.py [threshold] [mangled-prefix] [hipcc flags]"""
import collections, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gelato_amd", "csrc")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 128
prefix = sys.argv[2] if len(sys.argv) > 2 else "_ZN3gel11eval_kernelILb1ELb1ELb0ELb0EE"
out = "/tmp/vgpr_map.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-mllvm",
                "-disable-machine-licm", "-gline-tables-only", "--cuda-device-only", "-S", "gel_kernels.hip", "-o", out,
                "-DGEL_MIN_WAVES_PER_SIMD=2"] + sys.argv[3:], cwd=root, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[int(m.group(1))] = m.group(2)
start = [i for i, l in enumerate(lines) if l.startswith(prefix) and ": ;" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = None
hits = collections.Counter(); top = 0; order = []
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
    if not re.match(r"\s+[vdsgb]", l): continue
    regs = [int(x) for x in re.findall(r"\bv(\d+)\b", l)] + [int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", l)]
    if regs:
        top = max(top, max(regs))
        if max(regs) >= thr:
            if cur not in hits: order.append(cur)
            hits[cur] += 1
print("highest VGPR used:", top)
for k in order:
    print("   %4d  %s:%d" % (hits[k], k[0], k[1]))
