#!/bin/bash
# On the GPU box: times every build/variants/libgel_*.so (or the named ones) with scan_batch.py, interleaved twice.
# Usage: run_variants.sh "<workloads>" [names...]     env: SCAN_B
cd $GRAFT_REPO_ROOT
WL=${1:-"mixed-6x64"}; shift
names="$@"; [ -z "$names" ] && names=$(ls build/variants/libgel_*.so | sed 's/.*libgel_//; s/\.so//')
for rep in 1 2; do for n in $names; do
  echo "== $n (pass $rep)"
  GELATO_AMD_LIB=$GRAFT_REPO_ROOT/build/variants/libgel_$n.so SCAN_B=${SCAN_B:-16384} python3 tools/scan_batch.py $WL 2>/dev/null | grep '"B"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-14s B=%-6d jac=%d  %.4f ms  %.0f GB/s' % (d['workload'], d['B'], d['jac'], d['ms'], d['GBps']))"
done; done
