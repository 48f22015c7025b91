python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/full_t.log
python bench.py > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err
cat gpurun_out/full_t.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_f.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('captions_per_s'), v.get('error'))
PY
