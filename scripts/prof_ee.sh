#!/bin/bash
# kernel trace of the EOS-biased (early-exit) SCST step; prints the GPU idle gaps
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_ee
rm -rf $out
rocprofv3 --kernel-trace -d $out -- python $GRAFT_REPO_ROOT/bench.py --eos-bias 12 --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $GRAFT_REPO_ROOT/gpurun_out/prof_ee.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_ee.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('early_exit_eos_biased'))"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob
db=glob.glob('gpurun_out/prof_ee/*/*results.db')[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(c.execute(f"select s.kernel_name, d.start, d.end, d.stream_id from {kt} d join {ks} s on d.kernel_id=s.id order by d.start"))
# last ~3 steps: find adam kernels as step delimiters
ad=[i for i,r in enumerate(rows) if 'adam2' in r[0]]
a,b=ad[-3],ad[-1]
seg=rows[a:b+1]
span=(seg[-1][2]-seg[0][1])/1e3; busy=sum(r[2]-r[1] for r in seg)/1e3
print('2 steps: span %.0f us busy %.0f us, %d kernels'%(span,busy,len(seg)))
for i in range(1,len(seg)):
    g=(seg[i][1]-seg[i-1][2])/1e3
    if g>40: print(round(g), seg[i-1][0][:60], '->', seg[i][0][:60])
PY
