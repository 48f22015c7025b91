#!/bin/bash
# r5 evidence run (one gpurun call): GPU suite + smoke + the driver's bench command, then the profiles this round's numbers cite.
out=gpurun_out/r5z; mkdir -p $out; cd /root/repo; export R=/root/repo
timeout 600 python -m pytest tests -m gpu -q > $out/suite.log 2>&1; tail -3 $out/suite.log
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 600 $out/bench_n1.json; echo
scripts/prof_bench.sh r05_scst > $out/prof_scst.log 2>&1; tail -5 $out/prof_scst.log
scripts/pmc_passes.sh r05 > $out/pmc.log 2>&1; tail -4 $out/pmc.log
for c in "uxe updown_xe" "txe transformer_xe" "aoa aoa_nsc"; do set -- $c; scripts/prof_config.sh r05_$1 $2 > $out/prof_$1.log 2>&1; head -12 gpurun_out/r05_$1_kernel_stats.md | tail -7 | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_r05_attn
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05_attn -- python $R/scripts/tools_attn_large.py > $R/$out/attn_large.log 2>&1
cd $R; tail -1 $out/attn_large.log | cut -c1-400
db=$(find gpurun_out/prof_r05_attn -name "*.db" | head -1); python scripts/tools_prof.py $db 1 "rocprofv3 --kernel-trace --stats -- python scripts/tools_attn_large.py" > gpurun_out/r05_attention_large_batch_trace.md; grep attention_fwd gpurun_out/r05_attention_large_batch_trace.md | cut -c1-200
# keep the merged-back payload small (gpurun copies back at most 64 MiB): the summaries are what is kept
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -name "*.csv" -size +512k -delete 2>/dev/null; rm -rf gpurun_out/prof_r05_* ; du -sh gpurun_out | tail -1
