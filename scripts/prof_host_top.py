#!/usr/bin/env python3
"""cProfile of the TIMED steps of bench.py --config <cfg> (CAPMI_BENCH_CPROFILE): the functions the host spends its time in while
issuing a step, by own and by cumulative time.    python scripts/prof_host_top.py [config] [steps]"""
import os
import subprocess
import sys
cfg = sys.argv[1] if len(sys.argv) > 1 else 'aoa_nsc'
steps = sys.argv[2] if len(sys.argv) > 2 else '20'
env = dict(os.environ, CAPMI_BENCH_CPROFILE='1')
r = subprocess.run([sys.executable, 'bench.py', '--config', cfg, '--steps', steps, '--warmup', '2', '--no-cpu-baseline', '--no-prof', '--brief'],
                   env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
print('\n'.join(l for l in r.stderr.split('\n') if 'amdgpu.ids' not in l))
