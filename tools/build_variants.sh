#!/bin/bash
# Cross-compiles engine variants (extra -D switches) into build/variants/ HERE (no GPU needed); they travel to
# the GPU box with the snapshot (build/ is git-ignored, not gpurun-ignored).  Usage: build_variants.sh "name:-Dflags" ...
cd "$(dirname "$0")/../gelato_amd/csrc" || exit 1
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm $flags \
     -shared -o ../../build/variants/libgel_$name.so gel_kernels.hip gel_host.hip 2>/dev/null && echo "built $name [$flags]" || echo "FAILED $name" &
done
wait
