# the four BASELINE bench lines, one summary line each (box-to-box spread): tools/bench_lines.sh
cd $GRAFT_REPO_ROOT
for a in "" "--workload dense-6x64" "--workload stress-12x128 --batch 16384" "--workload 3x32 --residual-only --batch 65536"; do
python3 bench.py --no-cpu-baseline --no-extras $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['config']['workload'], 'value %.4g settled %.4g frac %.3f settled %.3f kernel_ms %.4f'%(d['value'], d.get('value_settled',0), r['frac'], r.get('frac_settled',0), r['kernel_ms']))"
done
