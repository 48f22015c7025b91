for o in 0 2 3 0 2 3; do CAPMI_ARES_OPT=$o python scripts/gemm_ablate.py 2>&1 | tail -1; done
for o in 2 3 2 3; do CAPMI_ARES_OPT=$o python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('opt=$o', d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac'])"; done
