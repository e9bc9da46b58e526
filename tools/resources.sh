#!/bin/bash
# (here, no GPU) VGPRs / scratch / occupancy of every eval_kernel instantiation: tools/resources.sh [extra hipcc flags]
cd "$(dirname "$0")/../gelato_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm "$@" \
  -Rpass-analysis=kernel-resource-usage -c gel_kernels.hip -o /tmp/gel_kernels_res.o 2>&1 | python3 -c '
import re, sys
name = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: name = m.group(1); vals = {}
    for key in ("VGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "VGPRs Spill", "SGPRs"):
        m = re.search(r"remark:\s+" + key + r": (\d+)", line)
        if m: vals[key] = m.group(1)
    if name and "LDS Size" in line:
        short = re.sub(r"_ZN3gel|EvNS_10ProblemDev.*", "", name)
        print("%-40s VGPRs %s scratch %s spill %s occupancy %s SGPRs %s" % (short, vals.get("VGPRs"), vals.get("ScratchSize \[bytes/lane\]"), vals.get("VGPRs Spill"), vals.get("Occupancy \[waves/SIMD\]"), vals.get("SGPRs")))
        name = None
'
