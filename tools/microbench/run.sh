#!/bin/bash
# on the GPU box: dynamic VALU instructions per lane-evaluation of every RHS piece
cd $GRAFT_REPO_ROOT/tools/microbench && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -o /tmp/instr_count instr_count.hip 2>&1 | grep -E "error" 
cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/ic -o ic -- /tmp/instr_count > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ic/**/*counter_collection.csv', recursive=True)[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d[r['Kernel_Name']][r['Counter_Name']] = d[r['Kernel_Name']].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
base = None
for k, v in d.items():
    per = v['SQ_INSTS_VALU'] / v['SQ_WAVES']
    if 'k_base' in k: base = per
for k, v in sorted(d.items(), key=lambda kv: kv[1]['SQ_INSTS_VALU']):
    w = v['SQ_WAVES']
    print("%-40s valu/wave %7.1f (net %7.1f)  salu %6.1f  lds %5.1f" % (k.split('(')[0], v['SQ_INSTS_VALU']/w, v['SQ_INSTS_VALU']/w - (base or 0), v['SQ_INSTS_SALU']/w, v['SQ_INSTS_LDS']/w))
PY
