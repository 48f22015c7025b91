#!/bin/bash
for v in 0 1 2 0 1 2; do
  CAPMI_ADAM_V=$v python bench.py --no-cpu-baseline --no-prof --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('ADAM_V=$v', d['value'], d['ms_per_step'])"
done
for ee in 0 4 0 4; do
  CAPMI_EARLY_EXIT=$ee python bench.py --no-cpu-baseline --no-prof --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('EARLY_EXIT=$ee', d['value'], d['ms_per_step'])"
done
python -m pytest tests/test_early_exit_gpu.py tests/test_kernels_gpu.py -x -q -k "early or adam" 2>&1 | tail -3
