#!/bin/bash
# rocprofv3 kernel trace of bench.py --config <cfg>; summary -> gpurun_out/<tag>_kernel_stats.md     usage: prof_config.sh tag cfg
tag=$1; cfg=$2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
db=$(find $out -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python scripts/tools_prof.py $db 14 "rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-prof ($tag)" > gpurun_out/${tag}_kernel_stats.md
head -60 gpurun_out/${tag}_kernel_stats.md
