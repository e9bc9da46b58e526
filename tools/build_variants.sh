#!/bin/bash
# Cross-compiles engine variants (extra -D switches) into build/variants/ HERE (no GPU needed); they travel to
# the GPU box with the snapshot (build/ is git-ignored, not gpurun-ignored).  Usage: build_variants.sh "name:-Dflags" ...
# A variant's old library is deleted first and its compiler log kept (build/variants/<name>.log), so that a failed build can
# neither be overlooked nor leave a stale library to be timed.
cd "$(dirname "$0")/../gelato_amd/csrc" || exit 1
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  rm -f ../../build/variants/libgel_$name.so
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp $flags \
     -shared -o ../../build/variants/libgel_$name.so gel_kernels.hip gel_host.hip > ../../build/variants/$name.log 2>&1 \
     && echo "built $name [$flags]" || { echo "FAILED $name (build/variants/$name.log):"; tail -5 ../../build/variants/$name.log; } ) &
done
wait
