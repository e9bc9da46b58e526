#!/bin/bash
# every BASELINE config on the record (bench line + rocprofv3 kernel stats + PMC traffic): tools/record_all.sh <tag>
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
tools/gpu_record.sh $TAG/mixed
tools/gpu_record.sh $TAG/dense --workload dense-6x64
tools/gpu_record.sh $TAG/stress --workload stress-12x128 --batch 16384
# a step of this launch is 0.39 ms: with the default W + K = 23 launches the timed region sits inside the start-up power transient, so more of both
tools/gpu_record.sh $TAG/3x32res --workload 3x32 --residual-only --batch 65536 --steps 200 --warmup 50
