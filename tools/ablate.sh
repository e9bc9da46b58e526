#!/bin/bash
# Builds ablation variants of the engine (profiling only) and times them. Run on the GPU box.
cd $GRAFT_REPO_ROOT/gelato_amd/csrc
for v in NOSTORE NODX NOPOS "NOSTORE -DGEL_ABL_NODX" ; do
  tag=$(echo $v | tr -d ' -' ); 
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -disable-machine-licm -DGEL_ABL_$v -shared -o /tmp/libgel_$tag.so gel_kernels.hip gel_host.hip 2>/dev/null
  echo "== variant $v"
  GELATO_AMD_LIB=/tmp/libgel_$tag.so python3 $GRAFT_REPO_ROOT/tools/scan_batch.py ${ABL_WORKLOAD:-dense-6x64} 2>/dev/null | grep -E '"B": (64|512|4096), "jac": true'
done
