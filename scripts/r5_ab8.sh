#!/bin/bash
# r5 A/B 8: deferred weight gradients on ONE stream with wide tiles allowed for them (new default) vs the side stream (CAPMI_DW_STREAM=1)
out=${1:-gpurun_out/r5o}; mkdir -p $out; cd /root/repo
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'), d['roofline'].get('avg_launch_us'))"; }
run() { name=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"; }
for rep in 1 2; do
run txe_one_stream.$rep transformer_xe CAPMI_DW_STREAM=0
run txe_side_stream.$rep transformer_xe CAPMI_DW_STREAM=1
run aoa_one_stream.$rep aoa_nsc CAPMI_DW_STREAM=0
run aoa_side_stream.$rep aoa_nsc CAPMI_DW_STREAM=1
done
run txe_one_stream_narrow transformer_xe CAPMI_DW_STREAM=0 CAPMI_X3_TILE=128
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_parity_gpu.py -q -x 2>&1 | tail -2
