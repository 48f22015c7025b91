#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 2>gpurun_out/r3g_bench.err | tail -1 > gpurun_out/r3g_bench.json
python -c "
import json;d=json.load(open('gpurun_out/r3g_bench.json'));print('SCST', d['value'], d['ms_per_step'], d['roofline']['frac'], d['early_exit_eos_biased'], d['cpu_baseline']['value'])" || tail -5 gpurun_out/r3g_bench.err
for c in updown_xe transformer_xe aoa_nsc newfc_xe; do
  timeout 900 python bench.py --config $c --steps 6 --warmup 2 2>gpurun_out/r3g_$c.err | tail -1 > gpurun_out/r3g_$c.json
  python -c "
import json;d=json.load(open('gpurun_out/r3g_$c.json'));print('$c', d['value'], d['ms_per_step'], d['roofline']['bound'], d['roofline']['frac'], d['cpu_baseline'])" || tail -5 gpurun_out/r3g_$c.err
done
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
