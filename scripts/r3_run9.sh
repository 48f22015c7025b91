#!/bin/bash
mkdir -p gpurun_out
{
echo "== unthrottled host, resident store, early exit OFF, 600 iterations"
MODES=1 CAPMI_EARLY_EXIT=0 CAPMI_TRAIN_LAG=-1 PYTHONFAULTHANDLER=1 timeout -s ABRT 150 bash scripts/train_e2e.sh 600 2>&1 | tail -60
echo "rc=$?"
} > gpurun_out/r3i_wedge.log 2>&1
cat gpurun_out/r3i_wedge.log
