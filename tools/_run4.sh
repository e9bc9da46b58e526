cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_aero_exact_fd.py tests/test_aero_engine.py -m gpu -q 2>&1 | tail -30
