cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "callback or aero or rows or driver or gn or shim or user" 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|Error|assert" | head -20
python3 tools/callback_loop.py 2>&1 | tail -3
python3 tools/cb_abi.py 2>&1 | tail -2 | cut -c1-700
