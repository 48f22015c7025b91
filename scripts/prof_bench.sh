#!/bin/bash
# rocprofv3 kernel trace of the bench command; summary -> gpurun_out/<tag>_kernel_stats.md     usage: prof_bench.sh tag [env...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
env "$@" rocprofv3 --kernel-trace --stats -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
db=$(find $out -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT
python scripts/tools_prof.py $db 15 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof ($tag; 8 init + 2 warm-up + 5 timed steps)" > gpurun_out/${tag}_kernel_stats.md
head -45 gpurun_out/${tag}_kernel_stats.md
