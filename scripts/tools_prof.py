#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (.db or *_kernel_stats.csv) as a markdown table: python tools_prof.py file [steps] [title]"""
import sqlite3
import sys


def main():
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    title = sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]
    if sys.argv[1].endswith('.csv'):      # rocprofv3 --output-format csv: *_kernel_stats.csv
        import csv
        rows = [(r['Name'], int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3,
                 float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3) for r in csv.DictReader(open(sys.argv[1]))]
        rows.sort(key=lambda r: -r[2])
    else:
        db = sqlite3.connect(sys.argv[1])
        rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                          "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print('# %s\n' % title)
    print('kernel time %.3f ms/step over %g steps, %.0f launches/step\n' % (tot / steps / 1e3, steps, sum(r[1] for r in rows) / steps))
    print('| kernel | calls/step | us/step | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|')
    for r in rows[:32]:
        print('| `%s` | %.1f | %.0f | %.2f | %.2f | %.2f | %.1f |' % (r[0][:100], r[1] / steps, r[2] / steps, r[3], r[4], r[5],
                                                                   100 * r[2] / tot))


if __name__ == '__main__':
    main()
