cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | head -20
python3 tools/other_kernels.py 2>&1 | tail -1 | cut -c1-700
python3 tools/cb_abi.py 2>&1 | tail -2 | cut -c1-600
