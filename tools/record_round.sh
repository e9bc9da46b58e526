# the whole record of a round: tools/record_all.sh + tools/record_others.sh, then one summary line per workload
cd $GRAFT_REPO_ROOT
tools/record_all.sh r03 > gpurun_out/record_r03.log 2>&1
tools/record_others.sh r03 > gpurun_out/others_r03.log 2>&1
tail -5 gpurun_out/record_r03.log | cut -c1-300
for w in mixed dense stress 3x32res; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/r03/$w/bench.json')); r=d['roofline']
print('$w', 'value %.4g settled %.4g frac %.3f settled %.3f kernel_ms %.4f bound %s'%(d['value'], d.get('value_settled',0), r['frac'], r.get('frac_settled',0), r['kernel_ms'], r['bound']))"; done
