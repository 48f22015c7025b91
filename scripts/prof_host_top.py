#!/usr/bin/env python3
"""cProfile of bench.py --config <cfg>: the functions the HOST spends its time in while issuing a step (cumulative and own time)"""
import cProfile, pstats, sys, io
cfg = sys.argv[1] if len(sys.argv) > 1 else 'aoa_nsc'
sys.argv = ['bench.py', '--config', cfg, '--steps', '30', '--warmup', '2', '--no-cpu-baseline', '--no-prof', '--brief']
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open('bench.py').read(), 'bench.py', 'exec'), {'__name__': '__main__', '__file__': 'bench.py'})
except SystemExit:
    pass
pr.disable()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print('\n'.join(l[:150] for l in s.getvalue().split('\n') if l.strip())[:5500])
