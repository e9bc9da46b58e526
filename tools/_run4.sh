cd $GRAFT_REPO_ROOT
tools/pmc_kernels.sh "" 2>&1 | grep -A40 "== eval_kernel" | head -60
