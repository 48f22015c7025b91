#!/bin/bash
# final measurement set of round 3: kernel stats, PMC passes, the default bench line, the --config lines
mkdir -p gpurun_out
bash scripts/prof_bench.sh r03f > /dev/null 2>&1
head -12 gpurun_out/r03f_kernel_stats.md
bash scripts/pmc_passes.sh r03f 2>&1 | tail -4
cp gpurun_out/r03f_pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null    # bench.py reads roofline.traffic from here
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03f_bench_n1.json
python -c "
import json;d=json.load(open('gpurun_out/r03f_bench_n1.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d.get('early_exit_eos_biased'))"
for c in updown_xe transformer_xe aoa_nsc newfc_xe; do
  python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03f_bench_$c.json
  python -c "
import json,sys;d=json.load(open('gpurun_out/r03f_bench_$c.json'));print('$c', d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'))"
done
