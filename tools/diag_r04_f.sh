#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python3 tools/cb_abi.py example mixed-6x64 2>/dev/null
python3 tools/b1_breakdown.py 2>/dev/null | tail -30
