#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, the bench line, and a rocprofv3 kernel-trace summary.
# Usage: tools/gpu_round.sh <tag> [bench args...]
TAG=${1:-r01}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 900 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras "$@" > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err )
find $OUT/prof -name "*stats*" | head; 
tail -3 $OUT/pytest_gpu.log; cat $OUT/bench.json; tail -2 $OUT/bench.err
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -8 $f; done
