#!/bin/bash
# kernel durations of the gate GEMM under ablations (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
export CAPMI_GEMM_PATH=${CAPMI_GEMM_PATH:-2}
for ab in "$@"; do
  rm -rf /tmp/prof_ab
  CAPMI_GEMM_ABLATE=$ab rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab --output-format csv -- python $GRAFT_REPO_ROOT/tools_gemm_one.py 60 0 > /dev/null 2>&1
  f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "ABLATE=$ab"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'gemm' in n: print('   %-60s calls=%s avg=%.2f us min=%.2f' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
