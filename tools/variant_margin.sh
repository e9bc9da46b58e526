#!/bin/bash
# like variant.sh but prints parity margins of the variant too
cd $GRAFT_REPO_ROOT/gelato_amd/csrc
tag=$(echo "$1" | tr -cd 'A-Za-z0-9_=' )
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm $1 -shared -o /tmp/libgel_$tag.so gel_kernels.hip gel_host.hip 2>/dev/null || { echo "build failed: $1"; exit 1; }
echo "== variant [$1]"
GELATO_AMD_LIB=/tmp/libgel_$tag.so python3 $GRAFT_REPO_ROOT/tests/parity_margin.py 2>/dev/null
GELATO_AMD_LIB=/tmp/libgel_$tag.so python3 $GRAFT_REPO_ROOT/tools/scan_batch.py mixed-6x64 2>/dev/null | grep -E '"jac": true' | sed 's/"evals_per_s.*//'
