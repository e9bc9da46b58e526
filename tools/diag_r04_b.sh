#!/bin/bash
# round-4 call B: GPU suite on the restructured front part + packed shards + downrange rows; A/B of the front variants; stamps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for rep in 1 2; do for v in main r3front apf1 apf4; do
  L=$PWD/build/variants/libgel_$v.so; [ $v = main ] && L=$PWD/gelato_amd/libgelato_amd.so
  echo "== $v (pass $rep)"
  GELATO_AMD_LIB=$L SCAN_B=16384,65536 timeout 300 python3 tools/scan_batch.py mixed-6x64 2>/dev/null | grep '"jac": true' | cut -c1-120
  GELATO_AMD_LIB=$L SCAN_B=65536 timeout 300 python3 tools/scan_batch.py dense-6x64 2>/dev/null | grep '"jac": true' | cut -c1-120
done; done
GELATO_AMD_LIB=$PWD/build/variants/libgel_stamp.so timeout 300 python3 tools/stamp_phases.py mixed-6x64 16384 2>&1 | tail -9
SCAN_B=16384 timeout 300 python3 tools/scan_batch.py stress-12x128 2>/dev/null | grep '"B"' | cut -c1-120
SCAN_B=65536 timeout 300 python3 tools/scan_batch.py 3x32 2>/dev/null | grep '"B"' | cut -c1-120
