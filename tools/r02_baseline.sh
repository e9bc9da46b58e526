#!/bin/bash
# round-2 opening call: GPU tests, then every BASELINE config on the record with the round-1 kernel
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r02a
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest_gpu.log
tail -3 gpurun_out/r02a/pytest_gpu.log
tools/gpu_record.sh r02a/mixed
tools/gpu_record.sh r02a/dense --workload dense-6x64
tools/gpu_record.sh r02a/stress --workload stress-12x128 --batch 4096
tools/gpu_record.sh r02a/3x32res --workload 3x32 --residual-only --batch 65536
