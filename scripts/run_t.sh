cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_g; rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof --no-other-configs > /dev/null 2>&1
db=$(find $out -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/scripts/prof_gaps.py $db 3 30 > $GRAFT_REPO_ROOT/gpurun_out/gaps_scst.log
rm -rf $out
cd $GRAFT_REPO_ROOT; cat gpurun_out/gaps_scst.log
