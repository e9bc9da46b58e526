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
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
echo "== $TAG"; cat $OUT/bench.json; tail -1 $OUT/bench.err; head -4 $OUT/kernel_stats.csv; cat $OUT/traffic.json
