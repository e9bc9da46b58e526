cd $GRAFT_REPO_ROOT
SCAN_B=16384 tools/run_variants.sh "mixed-6x64 dense-6x64" 2>&1 | grep -v "jac=0" | tail -40
GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_latalg.so timeout 600 python3 tests/parity_margin.py 2>&1 | tail -6
