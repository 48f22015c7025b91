#!/bin/bash
# the round's verification in one gpurun call: GPU suite, smoke(), the default bench line (-> gpurun_out/)
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/verify_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/verify_smoke.log 2>&1
python bench.py > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err
cat gpurun_out/verify_tests.log; tail -1 gpurun_out/verify_smoke.log
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/verify_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k, v in d['other_configs'].items():
    print(k, v.get('ms_per_step'), v.get('captions_per_s'), v.get('error'))
PY
