#!/bin/bash
# everything of round 4 on the record in one call (GPU box): tools/record_r04.sh [tag]
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
tools/record_all.sh $TAG > gpurun_out/$TAG/record_all.log 2>&1
tools/record_others.sh $TAG > gpurun_out/$TAG/record_others.log 2>&1
python3 tools/power_clock.py > gpurun_out/$TAG/power_clock.json 2> gpurun_out/$TAG/power_clock.err
python3 tools/batch_scan_record.py mixed-6x64 > gpurun_out/$TAG/batch_scan.json 2> gpurun_out/$TAG/batch_scan.err
python3 tools/shard_step.py mixed-6x64 > gpurun_out/$TAG/shard_step.json 2> gpurun_out/$TAG/shard_step.err
python3 tests/parity_margin.py > gpurun_out/$TAG/parity_margins.json 2> gpurun_out/$TAG/parity_margins.err
python3 tools/cb_abi.py example mixed-6x64 > gpurun_out/$TAG/callback_b1.jsonl 2>/dev/null
GELATO_AMD_LIB=$PWD/build/variants/libgel_stamp.so python3 tools/stamp_phases.py mixed-6x64 16384 > gpurun_out/$TAG/stamps_mixed.txt 2>/dev/null
bash tools/pmc_memsys.sh $TAG "--batch 16384" > gpurun_out/$TAG/memsys_mixed_B16384.txt 2>&1
python3 bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
tail -3 gpurun_out/$TAG/record_all.log; cat gpurun_out/$TAG/bench_default.json | cut -c1-600
