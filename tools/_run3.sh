cd $GRAFT_REPO_ROOT
SCAN_B=16384 timeout 900 tools/run_variants.sh "mixed-6x64 dense-6x64" main pf512 pf1024 pf2048 2>&1 | grep -v "jac=0" | tail -40
GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_stamp_pf.so timeout 300 python3 tools/stamp_phases.py mixed-6x64 16384 2>&1 | tail -9
