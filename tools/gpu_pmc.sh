#!/bin/bash
# PMC passes for the bench kernel (each counter group in its own rocprofv3 run; no tracing domains mixed in).
# Usage: tools/gpu_pmc.sh <tag> [bench args...]
TAG=${1:-r01}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name rc=$?"
}
BENCH_ARGS="$@"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
