#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3f_pytest.log
cat gpurun_out/r3f_pytest.log
bash scripts/prof_bench.sh r03a > /dev/null 2>&1
head -42 gpurun_out/r03a_kernel_stats.md
python bench.py 2>/dev/null | tail -1 > gpurun_out/r3f_bench.json; cat gpurun_out/r3f_bench.json
