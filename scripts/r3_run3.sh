#!/bin/bash
mkdir -p gpurun_out
{
echo "== launch cost (default env)"; ./scripts/ubench/launch_cost
echo "== launch cost HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 ./scripts/ubench/launch_cost
echo "== launch cost HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 ./scripts/ubench/launch_cost
for cfg in "0 2" "32 0" "32 2" "32 3" "34 2" "40 2" "33 2"; do
  set -- $cfg
  CAPMI_APL_ABLATE=$1 CAPMI_APL_PF=$2 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -1
done
CAPMI_APL_ABLATE=16 CAPMI_APL_PF=2 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -18
CAPMI_APL_ABLATE=48 CAPMI_APL_PF=2 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -18
CAPMI_APL_ABLATE=48 CAPMI_APL_PF=3 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -18
for kv in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$kv CAPMI_APL=0 timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('DEV_KERNARG=$kv APL=0', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline']['frac'])"
done
} > gpurun_out/r3c.log 2>&1
cat gpurun_out/r3c.log
