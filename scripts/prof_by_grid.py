#!/usr/bin/env python3
"""Per (kernel, grid, workgroup, LDS) averages of a rocprofv3 kernel-trace .db: python prof_by_grid.py file.db substring [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
pick = [c for c in ('grid_x', 'grid_y', 'workgroup_x', 'lds_size', 'grid_size_x', 'grid_size_y', 'workgroup_size_x', 'lds_block_size') if c in cols]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
q = ("select substr(name,1,60), %s, count(*), avg(end-start)/1e3, sum(end-start)/1e3 from kernels where name like ? group by name, %s "
     "order by 1, sum(end-start) desc" % (', '.join(pick), ', '.join(pick)))
print('columns: name, %s, calls/step, avg us, us/step' % ', '.join(pick))
for r in db.execute(q, ('%' + sys.argv[2] + '%',)):
    print(' | '.join(str(x) for x in r[:-3]), '| %.1f | %.2f | %.0f' % (r[-3] / steps, r[-2], r[-1] / steps))
