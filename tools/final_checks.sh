#!/bin/bash
# (GPU box) what the driver runs at round end, plus the one-rank torch.distributed path of bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_style.json 2> gpurun_out/bench_driver_style.err; echo "bench rc=$?"
python3 -c "
import json; d = json.load(open('gpurun_out/bench_driver_style.json')); r = d['roofline']
print('value %.2fM frac %.3f traffic %s (%s) bound %s (%s) fp64 busy %s warm %d cpu %.0f b1 %.4f' % (d['value']/1e6, r['frac'], r['traffic'], (r['traffic_source'] or '')[:40], r['bound'], r['bound_source'][:50], (r['fp64'] or {}).get('pipe_busy'), d['warmup_steps_run'], d['cpu_baseline']['value'], d['b1_host_callback_ms']))"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload 3x32 --residual-only --no-extras > gpurun_out/bench_3x32res_driver_style.json 2>/dev/null
python3 -c "
import json; d = json.load(open('gpurun_out/bench_3x32res_driver_style.json')); r = d['roofline']
print('3x32 res-only, driver flags: value %.1fM frac %.3f warm %d traffic %s' % (d['value']/1e6, r['frac'], d['warmup_steps_run'], r['traffic']))"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --batch 4096 --no-extras > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist rc=$?"; tail -2 gpurun_out/bench_dist1.err
python3 -c "
import json
l = [x for x in open('gpurun_out/bench_dist1.json') if x.startswith('{')]
d = json.loads(l[-1]); print('dist one rank:', d['n_gpus'], d['collective_backend'], '%.2fM' % (d['value']/1e6), 'cpu_baseline' in d, d['roofline']['frac'])"
