#!/bin/bash
# Cross-compiles engine variants (extra -D switches) into build/variants/ HERE (no GPU needed); they travel to
# the GPU box with the snapshot (build/ is git-ignored, not gpurun-ignored).  Usage: build_variants.sh "name:-Dflags" ...
# A variant's old library is deleted first and its compiler log kept (build/variants/<name>.log), so that a failed build can
# neither be overlooked nor leave a stale library to be timed.  The translation units are compiled as the Makefile compiles them
# (gel_kernels_aero.hip with its own scheduling strategy); KFLAGS_AERO in the environment overrides that unit's -mllvm flags.
cd "$(dirname "$0")/../gelato_amd/csrc" || exit 1
mkdir -p ../../build/variants
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm"
KA="${KFLAGS_AERO-}"
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  rm -f ../../build/variants/libgel_$name.so
  ( d=$(mktemp -d /tmp/gelvar.XXXXXX)
    { /opt/rocm/bin/hipcc $BASE -mllvm -amdgpu-sched-strategy=max-ilp $flags -c -o $d/k.o gel_kernels.hip \
      && /opt/rocm/bin/hipcc $BASE $KA $flags -c -o $d/a.o gel_kernels_aero.hip \
      && /opt/rocm/bin/hipcc $BASE -mllvm -amdgpu-sched-strategy=max-ilp $flags -c -o $d/h.o gel_host.hip \
      && /opt/rocm/bin/hipcc $BASE -shared -o ../../build/variants/libgel_$name.so $d/k.o $d/a.o $d/h.o ; } > ../../build/variants/$name.log 2>&1 \
     && echo "built $name [$flags]" || { echo "FAILED $name (build/variants/$name.log):"; tail -5 ../../build/variants/$name.log; }
    rm -rf $d ) &
done
wait
