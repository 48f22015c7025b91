#!/bin/bash
# r5 A/B 4: the SCST step with dW_logit on the side stream (CAPMI_BWD_SIDE=1) and / or the logit layer's Adam under the loop
out=${1:-gpurun_out/r5g}; mkdir -p $out; cd /root/repo
ms() { python -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['ms_per_step'], d.get('loss'))"; }
for rep in 1 2 3; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  CAPMI_BWD_SIDE=$1 CAPMI_EARLY_ADAM=$2 timeout 200 python bench.py --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline > $out/scst_s$1_e$2.$rep.json 2> $out/scst_s$1_e$2.$rep.err
  ms $out/scst_s$1_e$2.$rep.json "scst side=$1 early=$2"
done; done
