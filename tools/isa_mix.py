"""Static instruction mix of every eval_kernel instantiation in a device-only assembly listing
(hipcc ... --cuda-device-only -S gel_kernels.hip -o /tmp/gk/main.s).  usage: python3 tools/isa_mix.py /tmp/gk/main.s"""
import re
import sys
from collections import Counter

src = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/gk/main.s").read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if l.startswith("_ZN3gel11eval_kernel")]
for i, name in starts:
    c, slow = Counter(), Counter()
    f64 = 0
    for l in src[i + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if not t or t[0] in ".;" or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            f64 += "f64" in op
            if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|ldexp|frexp|fract|rndne|trig|cvt)", op):
                slow[op] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
            if op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            if op == "s_barrier":
                c["barrier"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        else:
            c["other"] += 1
    tmpl = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    print("<JAC,MFMA,SPLIT,PACK>=%s" % ",".join(tmpl.groups()), dict(c), "f64", f64, dict(slow))
