#!/bin/bash
CAPMI_PRE_STREAM=1 bash scripts/prof_bench.sh r03pre CAPMI_PRE_STREAM=1 > /dev/null 2>&1
head -20 gpurun_out/r03pre_kernel_stats.md
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/prof_r03pre/*/*_results.db')[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print(cols)
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
# find a steady-state step: print 40 consecutive kernels from the middle
mid = len(rows) * 2 // 3
t0 = rows[mid][1]
for name, s, e, q in rows[mid:mid + 45]:
    short = name.split('(')[0].split('::')[-1][:34]
    print('%9.1f %7.1f q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, short))
PY
