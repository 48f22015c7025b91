#!/bin/bash
# r5 A/B 3: pipelined epilogue operands (gemm_x3_common.h) -- correctness of both tilings, the epilogue table, Transformer XE
out=${1:-gpurun_out/r5f}; mkdir -p $out; cd /root/repo
for t in 128 256; do CAPMI_X3_TILE=$t timeout 200 python scripts/tools_x3w_bench.py --check 2>&1 | tail -1; done
for t in 128 256; do CAPMI_X3_TILE=$t timeout 120 python scripts/tools_x3w_bench.py --short 2>&1 | grep -v amdgpu.ids | sed "s/^/tile=$t /" | tee -a $out/epi.log; done
timeout 120 python scripts/tools_epilogue_bench.py 2>&1 | grep -v amdgpu.ids | tee $out/epilogue_bench.log
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'))"; }
for rep in 1 2; do for cfg in transformer_xe updown_xe; do
  timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/$cfg.$rep.json 2> $out/$cfg.$rep.err; ms $out/$cfg.$rep.json "$cfg"
done; done
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_parity_gpu.py -q -x 2>&1 | tail -3
