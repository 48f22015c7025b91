#!/bin/bash
# r5 A/B 6: skinny products on the wide kernel with swapped operands (CAPMI_X3_SWAP): correctness, shapes, configurations
out=${1:-gpurun_out/r5j}; mkdir -p $out; cd /root/repo
for t in 0 256; do CAPMI_X3_TILE=$t timeout 200 python scripts/tools_x3w_bench.py --check 2>&1 | grep -v amdgpu | grep "skinny\|all shapes\|BAD\|Error\|error" | tail -16; done
for sw in 1 0; do CAPMI_X3_SWAP=$sw timeout 120 python scripts/tools_x3w_bench.py --short 2>&1 | grep -v amdgpu.ids | sed "s/^/swap=$sw /" | tee -a $out/swap.log | tail -5; done
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'))"; }
for rep in 1 2; do for sw in 1 0; do for cfg in updown_xe aoa_nsc transformer_xe; do
  CAPMI_X3_SWAP=$sw timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/${cfg}_swap$sw.$rep.json 2> $out/${cfg}_swap$sw.$rep.err; ms $out/${cfg}_swap$sw.$rep.json "$cfg swap=$sw"
done; done; done
timeout 400 python -m pytest tests/test_full_size_parity_gpu.py tests/test_model_api_gpu.py tests/test_updown_gpu.py -q -x 2>&1 | tail -3
