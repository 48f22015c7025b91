#!/bin/bash
# r5 A/B 2: Transformer / UpDown XE with the wide fat-GEMM tiles on the main stream only, with / without the deferred-gradient
# side stream; epilogue-operand prefetch of the wide kernel.   usage: scripts/r5_ab2.sh <outdir>
out=${1:-gpurun_out/r5e}; mkdir -p $out; cd /root/repo
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'))"; }
run() { # name cfg env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"
}
for rep in 1 2; do
run txe_t128_dw1.$rep transformer_xe CAPMI_X3_TILE=128 CAPMI_DW_STREAM=1
run txe_auto_dw1.$rep transformer_xe CAPMI_X3_TILE=0 CAPMI_DW_STREAM=1
run txe_auto_dw0.$rep transformer_xe CAPMI_X3_TILE=0 CAPMI_DW_STREAM=0
run txe_t128_dw0.$rep transformer_xe CAPMI_X3_TILE=128 CAPMI_DW_STREAM=0
done
run txe_auto_nsw4_dw1 transformer_xe CAPMI_X3_TILE=0 CAPMI_X3W_NSW=4 CAPMI_DW_STREAM=1
run uxe_t128 updown_xe CAPMI_X3_TILE=128
run uxe_auto updown_xe CAPMI_X3_TILE=0
run uxe_auto_bxt0 updown_xe CAPMI_X3_TILE=0 CAPMI_BATCHED_XT=0
run aoa_auto aoa_nsc CAPMI_X3_TILE=0
run aoa_t128 aoa_nsc CAPMI_X3_TILE=128
for pf in 1 0; do CAPMI_X3_TILE=256 CAPMI_X3W_PF=$pf timeout 120 python scripts/tools_x3w_bench.py --short 2>&1 | grep -v amdgpu.ids | sed "s/^/pf=$pf /" | tee -a $out/pf.log; done
