#!/usr/bin/env python3
"""(here, no GPU) where the compiler spills in one eval_kernel instantiation: scratch stores / loads by source line, from a
device-only assembly listing with line tables.  tools/spill_map.py [mangled-prefix] [extra hipcc flags...]"""
import collections, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gelato_amd", "csrc")
prefix = sys.argv[1] if len(sys.argv) > 1 else "_ZN3gel11eval_kernelILb1ELb1ELb0ELb0EE"
out = "/tmp/spill_map.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-mllvm",
                "-disable-machine-licm", "-gline-tables-only", "--cuda-device-only", "-S", "gel_kernels.hip", "-o", out] + sys.argv[2:],
               cwd=root, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[int(m.group(1))] = m.group(2)
start = [i for i, l in enumerate(lines) if l.startswith(prefix) and ": ;" in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur = None
st, ld = collections.Counter(), collections.Counter()
valu = 0
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
    if "scratch_store" in l: st[cur] += 1
    if "scratch_load" in l: ld[cur] += 1
    if re.match(r"\s+v_", l): valu += 1
print("instantiation", prefix, "static VALU", valu, "scratch stores", sum(st.values()), "loads", sum(ld.values()))
for name, c in (("stores", st), ("loads", ld)):
    print(name + ":")
    for k, v in sorted(c.items(), key=lambda x: -x[1])[:25]:
        print("   %3d  %s:%d" % (v, k[0], k[1]))
