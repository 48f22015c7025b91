#!/bin/bash
mkdir -p gpurun_out
{
echo "== unthrottled host, resident store, 600 iterations"
MODES=1 CAPMI_TRAIN_LAG=-1 PYTHONFAULTHANDLER=1 timeout -s USR1 240 bash scripts/train_e2e.sh 600 2>&1 | tail -30
echo "rc=$?"
echo "== throttled (default)"
MODES=1 timeout 240 bash scripts/train_e2e.sh 300 2>&1 | tail -3
} > gpurun_out/r3h_wedge.log 2>&1
cat gpurun_out/r3h_wedge.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for c in updown_xe transformer_xe newfc_xe; do
  timeout 900 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/r3g_$c.err | tail -1 > gpurun_out/r3g_$c.json
  python -c "
import json;d=json.load(open('gpurun_out/r3g_$c.json'));print('$c', d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['ms_per_step'])" || tail -5 gpurun_out/r3g_$c.err
done
