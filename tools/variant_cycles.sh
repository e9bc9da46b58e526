#!/bin/bash
# GRBM_GUI_ACTIVE (GPU cycles) and duration of the bench kernel for a build variant: separates "fewer cycles"
# from "higher clock".  Usage: tools/variant_cycles.sh "<-D flags>"
cd $GRAFT_REPO_ROOT/gelato_amd/csrc
tag=$(echo "$1" | tr -cd 'A-Za-z0-9_=' )
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm $1 -shared -o /tmp/libgel_$tag.so gel_kernels.hip gel_host.hip 2>/dev/null || { echo "build failed: $1"; exit 1; }
export TMPDIR=/tmp GELATO_AMD_LIB=/tmp/libgel_$tag.so
cd /tmp && rm -rf /tmp/vc_$tag
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/vc_$tag -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > /tmp/vc_$tag.json 2>/dev/null
echo "== variant [$1]"
python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/vc_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "eval_kernel" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in acc.items()}, "launches", len(next(iter(acc.values()))))
print("kernel_ms", json.loads(open("/tmp/vc_$tag.json").read().strip().splitlines()[-1])["roofline"]["kernel_ms"])
PY
