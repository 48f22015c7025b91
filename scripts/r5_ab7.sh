#!/bin/bash
# r5 A/B 7: wide tiles for the side-stream (deferred) weight-gradient GEMMs after all?  With 4 staging waves a wide workgroup is 12
# waves x 125 registers: kernels without LDS can co-reside.  Also the whole suite once more with each tiling forced.
out=${1:-gpurun_out/r5k}; mkdir -p $out; cd /root/repo
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'), d['roofline'].get('achieved'))"; }
run() { name=$1; cfg=$2; shift 2; env "$@" timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/$name.json 2> $out/$name.err; ms $out/$name.json "$name"; }
for rep in 1 2; do
run txe_default.$rep transformer_xe CAPMI_X3_WIDE_DEFER=0
run txe_wdefer_nsw4.$rep transformer_xe CAPMI_X3_WIDE_DEFER=1 CAPMI_X3W_NSW=4
run txe_wdefer_nsw8.$rep transformer_xe CAPMI_X3_WIDE_DEFER=1 CAPMI_X3W_NSW=8
run aoa_default.$rep aoa_nsc CAPMI_X3_WIDE_DEFER=0
run aoa_wdefer_nsw4.$rep aoa_nsc CAPMI_X3_WIDE_DEFER=1 CAPMI_X3W_NSW=4
done
for t in 128 256; do CAPMI_X3_TILE=$t timeout 600 python -m pytest tests -m gpu -q -x --deselect tests/test_switches_gpu.py 2>&1 | tail -2; done
