#!/bin/bash
# One workload on the record (run on the GPU box via gpurun): the bench line with cpu_baseline, the
# rocprofv3 --kernel-trace --stats summary of the same command, and the FETCH_SIZE / WRITE_SIZE PMC passes
# (each in its own rocprofv3 run) -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, traffic.json}
# Usage: tools/gpu_record.sh <tag> [bench args...]      e.g.  tools/gpu_record.sh r02_dense --workload dense-6x64
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_profiled.json 2> $OUT/prof.err
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
done
python3 $R/tools/pmc_traffic.py $OUT "$@" > $OUT/traffic.json
# fp64 work and datapath occupancy (the roofline that binds when the store stream does not): instruction counters; cycles + busy counters
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/pmc_fp64a -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/pmc_fp64a.err
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_fp64b -o pmc -- python3 $R/bench.py --steps 4 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/pmc_fp64b.err
python3 $R/tools/pmc_fp64.py $OUT "$@" > $OUT/fp64.json
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_fp64a $OUT/pmc_fp64b
echo "== $TAG"; cat $OUT/bench.json; tail -1 $OUT/bench.err; head -4 $OUT/kernel_stats.csv; cat $OUT/traffic.json $OUT/fp64.json
