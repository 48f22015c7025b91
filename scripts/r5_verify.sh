#!/bin/bash
# r5 verification of HEAD (one gpurun call): the whole GPU suite, smoke(), the driver's bench command
out=${1:-gpurun_out/r5v}; mkdir -p $out; cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/suite.log 2>&1; tail -3 $out/suite.log
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err; tail -c 400 $out/bench_n1.json; echo
python - <<PY
import json
d = json.loads(open('$out/bench_n1.json').read().strip().splitlines()[-1])
print('scst', d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('step_ms'))
for k, v in d.get('other_configs', {}).items():
    print(k, v.get('ms_per_step'), v.get('step_ms'))
PY
