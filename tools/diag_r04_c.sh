#!/bin/bash
# call C: GPU suite; clock / power of the chip under the fused launch (rocm-smi sampled while a loop of launches runs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "#### idle"; rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -iE "sclk|mclk|power|perf|cap" | head -12
rocm-smi --showmaxpower 2>&1 | grep -i power | head -3
for wl in mixed-6x64; do
  for lib in gelato_amd/libgelato_amd.so; do
    ( GELATO_AMD_LIB=$PWD/$lib SCAN_B=65536 SCAN_SETTLE_S=6 timeout 120 python3 tools/scan_batch.py $wl > /tmp/scan.log 2>&1 & )
    sleep 7
    for i in 1 2 3; do echo "#### under load ($wl) sample $i"; rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|mclk|fclk|power" | head -8; sleep 0.7; done
    wait; sleep 4; grep '"jac": true' /tmp/scan.log | cut -c1-120
  done
done
python3 tools/shard_step.py mixed-6x64 > gpurun_out/shard_step.json 2> gpurun_out/shard_step.err; tail -3 gpurun_out/shard_step.err; python3 -c "
import json; d = json.load(open('gpurun_out/shard_step.json'))
for r in d['rows']: print(r['world'], r['B'], r['kernel_us_max'], r['exchange_standin_copy_us'], r['unpack_us_optional'], r['step_wall_us_rank0_synchronised'], r['single_gpu_fused_launch_us'])"
