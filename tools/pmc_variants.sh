#!/bin/bash
# SQ counter passes for engine variants (build/variants/libgel_<name>.so; "main" = the in-tree library).
# Usage: pmc_variants.sh "<bench args>" name...      -> gpurun_out/pmcv/<name>.json and a table on stdout
R=$GRAFT_REPO_ROOT; BA="$1"; shift
export TMPDIR=/tmp
cd /tmp
for n in "$@"; do
  OUT=$R/gpurun_out/pmcv/$n; rm -rf $OUT; mkdir -p $OUT
  LIB=$R/build/variants/libgel_$n.so; [ "$n" = main ] && LIB=$R/gelato_amd/libgelato_amd.so
  run() { local name=$1; shift
    GELATO_AMD_LIB=$LIB timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BA > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err || echo "pass $name failed"; }
  run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE
  run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
  run sq3 SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32
  python3 $R/tools/pmc_summary.py $OUT > /dev/null
  cp $OUT/pmc_summary.json $R/gpurun_out/pmcv/$n.json
  rm -rf $OUT
done
python3 - "$@" <<'PY'
import json, sys, os
R = os.environ["GRAFT_REPO_ROOT"]
names = sys.argv[1:]
data = {n: json.load(open("%s/gpurun_out/pmcv/%s.json" % (R, n))) for n in names}
keys = sorted({k for d in data.values() for k in d if isinstance(d[k], dict)})
print("%-28s" % "counter (mean per launch)" + "".join("%14s" % n for n in names))
for k in keys:
    print("%-28s" % k + "".join("%14.4g" % data[n].get(k, {}).get("mean_per_launch", float("nan")) for n in names))
PY
