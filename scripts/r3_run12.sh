#!/bin/bash
timeout 600 python -m pytest tests/test_planes_gpu.py tests/test_early_exit_gpu.py tests/test_updown_gpu.py -x -q 2>&1 | tail -6
for pre in 1 0 1 0; do
  CAPMI_PRE_STREAM=$pre timeout 300 python bench.py --no-cpu-baseline --no-prof --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('PRE_STREAM=$pre', d['value'], d['ms_per_step'])"
done
