#!/bin/bash
mkdir -p gpurun_out
{
for cfg in "0 0" "0 2" "0 3" "0 4" "1 0" "2 0" "4 0" "8 0" "3 0" "9 0" "10 0" "11 0" "6 0" "14 0" "15 0" "1 3" "2 3" "8 3" "11 3"; do
  set -- $cfg
  CAPMI_APL_ABLATE=$1 CAPMI_APL_PF=$2 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -1
done
CAPMI_APL_ABLATE=16 CAPMI_APL_PF=0 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -20
CAPMI_APL_ABLATE=16 CAPMI_APL_PF=3 timeout 120 python scripts/gemm_pl_ablate.py 2>&1 | tail -20
echo "== old kernel"
for a in 0 1 2 4 8 3 15; do CAPMI_ARES_ABLATE=$a timeout 120 python scripts/gemm_ablate.py 2>&1 | tail -1; done
CAPMI_ARES_ABLATE=16 timeout 120 python scripts/gemm_trace.py 2>&1 | tail -22
} > gpurun_out/r3b_ablate.log 2>&1
cat gpurun_out/r3b_ablate.log
