#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel-trace .db: python prof_gaps.py file.db [min_gap_us] [top]
Prints the total idle time inside the traced window, its histogram, and the largest gaps with the kernels on either side."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = db.execute('select start, end, substr(name, 1, 70) from kernels order by start').fetchall()
# the training steps end with the Adam kernel: analyse the last `nsteps` complete Adam-to-Adam intervals
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
adam = [i for i, r in enumerate(rows) if 'adam' in r[2]]
if len(adam) > nsteps:
    rows = rows[adam[-nsteps - 1] + 1:adam[-1] + 1]
    print('last %d steps (Adam to Adam)' % nsteps)
else:
    rows = rows[len(rows) // 2:]
busy = sum(e - s for s, e, _ in rows) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e3
gaps = []
last_end = rows[0][1]
for i in range(1, len(rows)):
    g = (rows[i][0] - last_end) / 1e3
    if g > 0:
        gaps.append((g, rows[i - 1][2], rows[i][2]))
    last_end = max(last_end, rows[i][1])
idle = sum(g for g, _, _ in gaps)
print('window %.1f us, %d kernels, busy %.1f us (%.1f %%), idle %.1f us' % (span, len(rows), busy, 100 * busy / span, idle))
print('(kernels may overlap on several streams: busy is the sum of durations)')
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
    sel = [g for g, _, _ in gaps if lo <= g < hi]
    print('  gaps %4g..%-6g us: %5d  sum %9.1f us' % (lo, hi, len(sel), sum(sel)))
agg = {}
for g, a, b in gaps:
    if g >= min_gap:
        k = (a, b)
        agg[k] = (agg.get(k, (0, 0))[0] + g, agg.get(k, (0, 0))[1] + 1)
print('largest contributors (gap >= %g us), by total idle time:' % min_gap)
for (a, b), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print('  %8.1f us in %4d gaps | after %-70s | before %s' % (tot, n, a, b))
