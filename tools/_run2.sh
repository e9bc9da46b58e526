cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 600 python3 tests/parity_margin.py 2>&1 | tail -8
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
SCAN_B=16384 timeout 300 python3 tools/scan_batch.py mixed-6x64 dense-6x64 2>&1 | grep -v "^$" | tail -6
