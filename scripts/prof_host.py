#!/usr/bin/env python3
"""cProfile of bench.py --config <cfg>: who calls the blocking torch methods (.to / .clone / .item) inside the step loop"""
import cProfile, pstats, sys, io
sys.argv = ['bench.py', '--config', sys.argv[1] if len(sys.argv) > 1 else 'aoa_nsc', '--steps', '20', '--warmup', '2', '--no-cpu-baseline', '--no-prof']
pr = cProfile.Profile()
pr.enable()
try:
    exec(compile(open('bench.py').read(), 'bench.py', 'exec'), {'__name__': '__main__', '__file__': 'bench.py'})
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s)
ps.print_callers("method 'to' of", "method 'clone'", "method 'item'", "method 'tolist'", "synchronize")
out = s.getvalue()
print('\n'.join(l for l in out.split('\n') if l.strip())[:6000])
