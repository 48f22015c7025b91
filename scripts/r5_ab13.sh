#!/bin/bash
# r5 A/B 13: the reference cycle of the rollout Functions broken (activations freed with the step's graph, not by the cyclic
# collector): memory test of the three families, then the Transformer XE child five times (step_ms shows stalls)
out=${1:-gpurun_out/r5c1}; mkdir -p $out; cd /root/repo
timeout 300 python -m pytest tests/test_model_api_gpu.py -q -x -m gpu -k "frees_its_activations or raw_logit or golden" -p no:cacheprovider > $out/tests.log 2>&1; tail -3 $out/tests.log
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('step_ms'))"; }
run() { name=$1; shift; env "$@" > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"; }
for rep in 1 2 3 4; do
run txe.$rep timeout 100 python bench.py --config transformer_xe --steps 20 --warmup 5 --no-cpu-baseline
done
run aoa.1 timeout 100 python bench.py --config aoa_nsc --steps 20 --warmup 5 --no-cpu-baseline
