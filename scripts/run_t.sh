python bench.py > gpurun_out/bench_g.json 2> gpurun_out/bench_g.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_g.json') if l.startswith('{')][-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['other_configs'].items(): print(k, v.get('ms_per_step'), v.get('captions_per_s'), v.get('error'))
PY
bash scripts/prof_config.sh mh_txe transformer_xe > /dev/null 2>&1
bash scripts/prof_config.sh mh_aoa aoa_nsc > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_txe gpurun_out/prof_mh_aoa
