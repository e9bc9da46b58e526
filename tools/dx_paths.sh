#!/bin/bash
# D.X on the matrix pipe (GEL_FLAGS=1) against wavefront dot-products (GEL_FLAGS=2) by nodes per phase (GPU box)
cd $GRAFT_REPO_ROOT
for wl in 3x8 3x16 3x32 mixed-6x64; do for f in 1 2; do
  echo "== $wl flags=$f ($( [ $f = 1 ] && echo MFMA || echo VALU ))"
  GEL_FLAGS=$f SCAN_B=${SCAN_B:-32768} python3 tools/scan_batch.py $wl 2>/dev/null | grep '"B"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   jac=%d  %.4f ms  %.3g evals/s' % (d['jac'], d['ms'], d['evals_per_s']))"
done; done
