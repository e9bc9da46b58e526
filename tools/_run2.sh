cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError:" | head -40
SCAN_B=16384 timeout 300 python3 tools/scan_batch.py mixed-6x64 dense-6x64 2>&1 | grep -v "^$" | grep '"jac"'
SCAN_B=4096 timeout 300 python3 tools/scan_batch.py stress-12x128 2>&1 | grep -v "^$" | grep '"jac"'
SCAN_B=65536 timeout 300 python3 tools/scan_batch.py 3x32 2>&1 | grep -v "^$" | grep '"jac"'
