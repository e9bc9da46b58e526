cd $GRAFT_REPO_ROOT
SCAN_B=65536 timeout 900 tools/run_variants.sh "3x32" main res6 res6np 2>&1 | grep "jac=0" | tail -40
