#!/bin/bash
# round-3 GPU call 1: planes path correctness first, then micro-benchmarks, then the full GPU suite and the bench A/B
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_planes_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3a_planes_test.log
cat gpurun_out/r3a_planes_test.log
{
for nt in 0 1; do for opt in 1 0; do
  echo "== CAPMI_APL_NT=$nt CAPMI_APL_OPT=$opt"
  CAPMI_APL_NT=$nt CAPMI_APL_OPT=$opt timeout 300 python scripts/tools_gemm_pl.py 60 2>&1 | tail -2
done; done
} > gpurun_out/r3a_gemm_pl.log 2>&1
cat gpurun_out/r3a_gemm_pl.log
for apl in 1 0 1 0; do
  CAPMI_APL=$apl timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3a_bench_apl$apl.json
  python -c "
import json;d=json.load(open('gpurun_out/r3a_bench_apl$apl.json'));print('APL=$apl', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline']['frac'])"
done 2>&1 | tee gpurun_out/r3a_bench_ab.log
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a_pytest.log
cat gpurun_out/r3a_pytest.log
