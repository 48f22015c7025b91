#!/bin/bash
# rocprofv3 kernel trace of any python command: prof_any.sh tag steps script [args...]  -> gpurun_out/<tag>_kernel_stats.md
tag=$1; steps=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $out -- python "$@" > gpurun_out/prof_$tag.log 2>&1
tail -3 gpurun_out/prof_$tag.log
db=$(find $out -name "*.db" | head -1)
python scripts/tools_prof.py $db $steps "rocprofv3 --kernel-trace --stats -- python $*" > gpurun_out/${tag}_kernel_stats.md
head -36 gpurun_out/${tag}_kernel_stats.md
