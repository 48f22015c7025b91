#!/bin/bash
# r5 A/B 11: AoA decode step with the slab consumers (CAPMI_AOA_SLABS=1, default) against the separate reduce / split launches
out=${1:-gpurun_out/r5s}; mkdir -p $out; cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_api_gpu.py tests/test_aoa_train_mode_gpu.py -q -x -m gpu -k "slab or aoa or mha or layernorm or glu or select" -p no:cacheprovider > $out/tests.log 2>&1; tail -3 $out/tests.log
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('step_ms'))"; }
run() { name=$1; shift; env "$@" > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"; }
for rep in 1 2 3; do
run slabs1.$rep CAPMI_AOA_SLABS=1 timeout 200 python bench.py --config aoa_nsc --steps 20 --warmup 5 --no-cpu-baseline --no-prof
run slabs0.$rep CAPMI_AOA_SLABS=0 timeout 200 python bench.py --config aoa_nsc --steps 20 --warmup 5 --no-cpu-baseline --no-prof
done
