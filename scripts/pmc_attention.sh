#!/bin/bash
# FETCH_SIZE of the fused attention kernel with / without the XCD-sorted row map (variants build)
export CAPMI_LIB=$GRAFT_REPO_ROOT/variants/libcapmi.so
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  rm -rf $R/gpurun_out/pmc_att_$x
  CAPMI_ATT_XCD=$x timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_att_$x -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_att_$x.log 2>&1
  python - <<PY
import csv, glob
v = {}
for f in glob.glob('$R/gpurun_out/pmc_att_$x/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and ('attention_fwd_v2' in r['Kernel_Name'] or 'attention_bwd_v2' in r['Kernel_Name']):
            v.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append(float(r['Counter_Value']))
for k, a in v.items():
    print('CAPMI_ATT_XCD=$x', k, 'launches', len(a), 'FETCH_SIZE per launch (KiB x 2048 B, gfx950 correction): %.2f MB' % (sum(a) / len(a) * 2048 / 1e6))
PY
  find $R/gpurun_out/pmc_att_$x -name "*.csv" -size +5M -delete
done
