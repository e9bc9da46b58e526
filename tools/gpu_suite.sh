#!/bin/bash
# (GPU box) the -m gpu suite; tools/gpu_suite.sh [pytest args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -15
