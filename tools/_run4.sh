cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "aero or callback or rows or driver or gn or shim" 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|assert|^E " | head -20
python3 tools/other_kernels.py 2>&1 | tail -1 | cut -c1-500
python3 tools/cb_abi.py 2>&1 | tail -2 | cut -c1-600
