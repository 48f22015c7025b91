#!/bin/bash
# r5 A/B 9: where does a slow Transformer XE run lose its time?  per-step spread (bench `step_ms`), with and without the CPU
# baseline / HIP-event sampling, both stream modes, several repetitions in one call
out=${1:-gpurun_out/r5p}; mkdir -p $out; cd /root/repo
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('step_ms'), d['roofline'].get('avg_launch_us'))"; }
run() { name=$1; shift; env "$@" > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"; }
for rep in 1 2 3; do
run brief.$rep timeout 200 python bench.py --config transformer_xe --steps 8 --warmup 3 --brief
run brief_nocpu.$rep timeout 200 python bench.py --config transformer_xe --steps 8 --warmup 3 --brief --no-cpu-baseline
run brief_noprof.$rep timeout 200 python bench.py --config transformer_xe --steps 8 --warmup 3 --brief --no-cpu-baseline --no-prof
run side.$rep CAPMI_DW_STREAM=1 timeout 200 python bench.py --config transformer_xe --steps 8 --warmup 3 --brief --no-cpu-baseline
run long.$rep timeout 200 python bench.py --config transformer_xe --steps 30 --warmup 5 --no-cpu-baseline
done
