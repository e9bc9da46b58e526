#!/bin/bash
# everything of round 5 on the record in one call (GPU box): tools/record_r05.sh [tag]   (then, here: tools/collect_profiles.sh <tag>)
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
tools/record_all.sh $TAG > gpurun_out/$TAG/record_all.log 2>&1                      # the four BASELINE workloads: bench line, rocprofv3 stats, PMC traffic, fp64
tools/record_others.sh $TAG > gpurun_out/$TAG/record_others.log 2>&1                # aero / expand / update / rows / jac_fd kernels
bash tools/pmc_instmix.sh "" > gpurun_out/$TAG/instruction_mix.txt 2>&1             # dynamic instruction mix of the fused kernel per wavefront
python3 tools/power_clock.py > gpurun_out/$TAG/power_clock.json 2> gpurun_out/$TAG/power_clock.err
python3 tests/parity_margin.py > gpurun_out/$TAG/parity_margins.json 2> gpurun_out/$TAG/parity_margins.err
python3 tools/cb_abi.py example mixed-6x64 > gpurun_out/$TAG/callback_b1.jsonl 2>/dev/null        # one callback through the C-ABI
python3 tools/callback_loop.py > gpurun_out/$TAG/callback_python.jsonl 2>/dev/null               # objfunc / sens through the reference-named functions
python3 tools/b1_breakdown.py mixed-6x64 > gpurun_out/$TAG/b1_breakdown.json 2>/dev/null
GEL_NO_COO_DIRECT=1 python3 tools/b1_breakdown.py mixed-6x64 > gpurun_out/$TAG/b1_breakdown_compact_path.json 2>/dev/null
python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
tail -3 gpurun_out/$TAG/record_all.log; cat gpurun_out/$TAG/bench_default.json | cut -c1-600
