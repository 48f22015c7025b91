#!/bin/bash
# rocprofv3 kernel trace of the secondary XE workloads: prof_xe.sh tag which
tag=$1; which=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/profxe_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $out -- python scripts/tools_xe_bench.py $which > gpurun_out/profxe_$tag.log 2>&1
tail -2 gpurun_out/profxe_$tag.log
db=$(find $out -name "*.db" | head -1)
python scripts/tools_prof.py $db 17 "rocprofv3 --kernel-trace --stats -- python scripts/tools_xe_bench.py $which (4 warm-up + 10 timed XE steps + 3 greedy decodes)" > gpurun_out/${tag}_xe_kernel_stats.md
head -40 gpurun_out/${tag}_xe_kernel_stats.md
