#!/bin/bash
# (here, no GPU) Where does an instantiation of the fused kernel spill?  Compiles gel_kernels.hip with line tables and lists every
# scratch_load / scratch_store of the chosen kernel with the source line it belongs to.
# usage: tools/spill_map.sh <mangled-name prefix> [extra -D flags]     e.g. _ZN3gel11eval_kernelILb1ELb1ELb0ELb0ELb1ELb1ELb1E
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
K="$1"; shift
D=$(mktemp -d /tmp/spillmap.XXXXXX)
( cd "$D" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm \
    -mllvm -amdgpu-sched-strategy=max-ilp -gline-tables-only -save-temps "$@" -c "$ROOT/gelato_amd/csrc/gel_kernels.hip" -o k.o 2>/dev/null )
python3 - "$D/gel_kernels-hip-amdgcn-amd-amdhsa-gfx950.s" "$K" <<'EOF'
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
cur = None
n = 0
for i in range(start, len(lines)):
    l = lines[i]
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (m.group(1), m.group(2))
    if 'scratch_' in l:
        print(i - start, cur, l.strip()[:100])
    if re.match(r'\s+v_|\s+s_|\s+ds_|\s+buffer_|\s+global_', l):
        n += 1
    if l.startswith('.Lfunc_end'):
        break
print("instructions:", n)
EOF
echo "asm kept in $D"
