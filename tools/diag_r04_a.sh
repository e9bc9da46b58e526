#!/bin/bash
# round-4 diagnosis call: where the front part of a wavefront's life goes (stamps with / without the store stream) and
# what the memory system says (L1 request latencies, TLB, queue-full cycles)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in stamp stamp_nostore stamp_same; do
  echo "#### $v"; GELATO_AMD_LIB=$PWD/build/variants/libgel_$v.so timeout 300 python3 tools/stamp_phases.py mixed-6x64 16384 2>&1 | tail -9
done
bash tools/run_variants.sh "mixed-6x64" nostore sameaddr 2>&1 | tail -12
SCAN_B=16384,65536 timeout 300 python3 tools/scan_batch.py mixed-6x64 2>&1 | grep '"jac": true'
bash tools/pmc_memsys.sh base "--batch 16384" 2>&1 | tail -80
GELATO_AMD_LIB=$PWD/build/variants/libgel_sameaddr.so bash tools/pmc_memsys.sh sameaddr "--batch 16384" 2>&1 | tail -80
