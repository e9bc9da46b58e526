#!/bin/bash
# (here, no GPU) registers / scratch / LDS / occupancy of every kernel of gel_kernels.hip.  Usage: tools/resources.sh [extra -D flags]
cd "$(dirname "$0")/../gelato_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
  -Rpass-analysis=kernel-resource-usage -c gel_kernels.hip -o /tmp/gel_kernels_res.o 2>&1 | python3 -c "
import sys, re
cur = None
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = m.group(1); rec = {}
        continue
    m = re.search(r'remark: +(\w[\w /\[\]-]*): (\S+)', l)
    if m and cur:
        rec[m.group(1).strip()] = m.group(2)
        if m.group(1).startswith('LDS Size'):
            import subprocess
            name = subprocess.run(['c++filt', cur], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(.*', '', name)
            print('%-70s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-4s occ %-2s LDS %s' % (name[:70], rec.get('VGPRs'), rec.get('AGPRs'), rec.get('TotalSGPRs'), rec.get('ScratchSize [bytes/lane]'), rec.get('Occupancy [waves/SIMD]'), rec.get('LDS Size [bytes/block]')))
"
