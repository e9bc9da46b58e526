#!/usr/bin/env python3
"""Static VALU mix per source line of one kernel, from an assembly listing with line tables
(hipcc ... -gline-tables-only --cuda-device-only -S gel_kernels.hip -o /tmp/gk2/main_g.s).
usage: tools/line_mix.py /tmp/gk2/main_g.s <mangled-kernel-prefix> [top]"""
import re, sys
from collections import Counter, defaultdict
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else "_ZN3gel11eval_kernelILb1ELb1ELb0ELb0EE"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
for l in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
i = [k for k, l in enumerate(src) if l.startswith(want) and ": " in l][0]
cur = None
mix = defaultdict(Counter)
for l in src[i + 1:]:
    if l.startswith(".Lfunc_end"): break
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if not t or t[0] in ".;" or t.endswith(":"): continue
    op = t.split()[0]
    if not op.startswith("v_"): continue
    if op.startswith("v_mfma"): k = "mfma"
    elif op.startswith("v_mov"):
        s = t.split(",")[1].strip().split()[0]
        k = "mov_v" if s.startswith(("v", "a")) else "mov_c"
    elif op.startswith("v_cndmask"): k = "cnd"
    elif op.startswith("v_cmp"): k = "cmp"
    elif "f64" in op: k = "f64"
    else: k = "int"
    mix[cur][k] += 1
tot = Counter()
for c in mix.values(): tot.update(c)
print("kernel total", dict(tot))
rows = sorted(mix.items(), key=lambda kv: -(sum(kv[1].values()) - kv[1]["f64"] - kv[1]["mfma"]))
for (f, ln), c in rows[:top]:
    print("%-22s %5d  f64 %4d  mov_c %3d mov_v %3d cnd %3d cmp %3d int %3d" % (f, ln, c["f64"], c["mov_c"], c["mov_v"], c["cnd"], c["cmp"], c["int"]))
