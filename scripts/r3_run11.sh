#!/bin/bash
mkdir -p gpurun_out
bash scripts/prof_bench.sh r03b > /dev/null 2>&1
head -44 gpurun_out/r03b_kernel_stats.md
bash scripts/pmc_passes.sh r03 2>&1 | tail -6
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03b_bench_n1.json
python -c "
import json;d=json.load(open('gpurun_out/r03b_bench_n1.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['traffic'], d['early_exit_eos_biased'])"
