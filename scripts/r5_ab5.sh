#!/bin/bash
# r5 A/B 5: swizzled LDS rows in the 128 x 128 kernel (vs the r5b table), AoA backward with capmi_split_halves; AoA / kernel tests
out=${1:-gpurun_out/r5i}; mkdir -p $out; cd /root/repo
CAPMI_X3_TILE=128 timeout 300 python scripts/tools_x3w_bench.py 2>&1 | grep -v "amdgpu.ids\|^edge" | tee $out/t128_swz.log | tail -24
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_aoa_train_mode_gpu.py tests/test_full_size_parity_gpu.py tests/test_sparse_logp_gpu.py -q -x 2>&1 | tail -3
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'))"; }
for rep in 1 2; do for cfg in aoa_nsc transformer_xe updown_xe newfc_xe; do
  timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/$cfg.$rep.json 2> $out/$cfg.$rep.err; ms $out/$cfg.$rep.json "$cfg"
done; done
timeout 200 python bench.py --steps 20 --warmup 3 --no-other-configs --no-cpu-baseline > $out/scst.json 2> $out/scst.err; ms $out/scst.json scst
