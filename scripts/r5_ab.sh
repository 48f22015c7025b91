#!/bin/bash
# r5 A/B inside one gpurun call: the GPU suite on the new defaults, then config-level timings of the switches
# (CAPMI_BWD_SIDE side-stream weight gradients, CAPMI_X3_TILE fat-GEMM tiling).  usage: scripts/r5_ab.sh <outdir>
out=${1:-gpurun_out/r5d}; mkdir -p $out; cd /root/repo
timeout 500 python -m pytest tests -m gpu -q -x > $out/suite.log 2>&1; tail -4 $out/suite.log
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'))"; }
for rep in 1 2; do for side in 0 1 2 3; do
  CAPMI_BWD_SIDE=$side timeout 200 python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline > $out/scst_side$side.$rep.json 2> $out/scst_side$side.$rep.err
  ms $out/scst_side$side.$rep.json "scst side=$side"
done; done
for cfg in updown_xe transformer_xe aoa_nsc newfc_xe; do for tile in 0 128; do
  CAPMI_X3_TILE=$tile timeout 200 python bench.py --config $cfg --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/${cfg}_tile$tile.json 2> $out/${cfg}_tile$tile.err
  ms $out/${cfg}_tile$tile.json "$cfg tile=$tile"
done; done
CAPMI_BWD_SIDE=0 timeout 200 python bench.py --config updown_xe --steps 8 --warmup 3 --brief --no-cpu-baseline > $out/updown_xe_side0.json 2> $out/updown_xe_side0.err; ms $out/updown_xe_side0.json "updown_xe side=0"
