cd $GRAFT_REPO_ROOT
RES_ONLY=1 GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_stamp.so timeout 300 python3 tools/stamp_phases.py 3x32 65536 2>&1 | tail -9
RES_ONLY=1 GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_stamp.so timeout 300 python3 tools/stamp_phases.py mixed-6x64 16384 2>&1 | tail -9
GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_stamp.so timeout 300 python3 tools/stamp_phases.py stress-12x128 4096 2>&1 | tail -9
tools/pmc_kernels.sh "--workload 3x32 --residual-only --batch 65536" 2>&1 | grep -A30 "== eval_kernel" | head -34
