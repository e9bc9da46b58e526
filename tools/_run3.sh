cd $GRAFT_REPO_ROOT
tools/record_others.sh r03 > gpurun_out/others_r03.log 2>&1
tail -30 gpurun_out/others_r03.log
export TMPDIR=/tmp; mkdir -p gpurun_out/cbtrace; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cbtrace/prof -o trace -- python3 $GRAFT_REPO_ROOT/tools/cb_abi.py mixed-6x64 > $GRAFT_REPO_ROOT/gpurun_out/cbtrace/cb.json 2>$GRAFT_REPO_ROOT/gpurun_out/cbtrace/err
cp $(find $GRAFT_REPO_ROOT/gpurun_out/cbtrace/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/cbtrace/kernel_stats.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cbtrace/prof
cat $GRAFT_REPO_ROOT/gpurun_out/cbtrace/kernel_stats.csv | cut -c1-200; cat $GRAFT_REPO_ROOT/gpurun_out/cbtrace/cb.json | cut -c1-800
