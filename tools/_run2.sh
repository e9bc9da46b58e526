cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | head -20
for f in 0 16; do echo "flags $f"; GEL_FLAGS=$f SCAN_B=16384 timeout 300 python3 tools/scan_batch.py mixed-6x64 dense-6x64 2>&1 | grep '"jac"' | cut -c1-110
GEL_FLAGS=$f SCAN_B=65536 timeout 300 python3 tools/scan_batch.py 3x32 2>&1 | grep '"jac"' | cut -c1-110; done
