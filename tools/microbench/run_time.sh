#!/bin/bash
# on the GPU box: time per call of every RHS piece at 4 waves/SIMD
cd $GRAFT_REPO_ROOT/tools/microbench && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm -o /tmp/piece_time piece_time.hip 2>&1 | grep -E "error"
/tmp/piece_time
