cd $GRAFT_REPO_ROOT
SCAN_B=16384 tools/run_variants.sh "mixed-6x64 dense-6x64" 2>&1 | grep -v "jac=0" | tail -30
tools/pmc_kernels.sh "" 2>&1 | grep -A2 "== eval_kernel" | head -20
