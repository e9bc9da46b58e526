#!/bin/bash
# everything of round 6 on the record in one call (GPU box): tools/record_r06.sh [tag]   (then, here: tools/collect_profiles.sh <tag>)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
tools/record_all.sh $TAG > gpurun_out/$TAG/record_all.log 2>&1                      # the four BASELINE workloads: bench line, rocprofv3 stats, PMC traffic, fp64
tools/record_others.sh $TAG > gpurun_out/$TAG/record_others.log 2>&1                # aero / expand / update / rows / jac_fd kernels
bash tools/pmc_instmix.sh "" > gpurun_out/$TAG/instruction_mix.txt 2>&1             # dynamic instruction mix of the fused kernel per wavefront
python3 tests/parity_margin.py > gpurun_out/$TAG/parity_margins.json 2> gpurun_out/$TAG/parity_margins.err
python3 tools/batch_scan.py > gpurun_out/$TAG/batch_scan.json 2> gpurun_out/$TAG/batch_scan.err
# defect groups + aero rows: the two kernels against the fused launch, HIP events and the per-kernel split of a kernel trace
GEL_AERO_FUSED=0 python3 tools/aero_fused_probe.py mixed-6x64 65536 20 > gpurun_out/$TAG/aero_fused_default.json 2>/dev/null
GEL_AERO_FUSED=1 python3 tools/aero_fused_probe.py mixed-6x64 65536 20 > gpurun_out/$TAG/aero_fused_fused.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && GEL_AERO_FUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/aero_fused_prof -o t -- python3 $GRAFT_REPO_ROOT/tools/aero_fused_probe.py mixed-6x64 65536 20 > /dev/null 2>&1 )
cp $(find gpurun_out/$TAG/aero_fused_prof -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/aero_fused_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/$TAG/aero_fused_prof
python3 tools/placement_vmm.py 1 mixed-6x64 65536 vmm_off0MB,vmm_off2MB,vmm_off3MB,vmm_off5MB,vmm_off7MB,vmm_off11MB,vmm_off13MB,vmm_off17MB > gpurun_out/$TAG/placement_alternation.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
tail -3 gpurun_out/$TAG/record_all.log; cat gpurun_out/$TAG/bench_default.json | cut -c1-600
