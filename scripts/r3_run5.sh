#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q 2>&1 | tail -5
for a in 0 2 8 10; do CAPMI_LC_ABLATE=$a timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -1; done
CAPMI_LC_ABLATE=16 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -9
timeout 300 python scripts/tools_gemm_pl.py 60 2>&1 | tail -2
for lc in 1 0 1; do
  CAPMI_LC=$lc timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('LC=$lc', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline']['frac'])"
done
} > gpurun_out/r3e.log 2>&1
cat gpurun_out/r3e.log
