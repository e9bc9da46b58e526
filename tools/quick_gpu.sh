#!/bin/bash
# GPU tests + batch scan of the four workloads (run on the GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
SCAN_B=${SCAN_B:-4096,16384} timeout 600 python3 tools/scan_batch.py mixed-6x64 dense-6x64 2>&1 | grep -v "^$" | tail -12
SCAN_B=4096 timeout 600 python3 tools/scan_batch.py stress-12x128 2>&1 | tail -2
SCAN_B=65536 timeout 600 python3 tools/scan_batch.py 3x32 2>&1 | tail -2
