#!/bin/bash
# PMC passes over the fat-GEMM micro-benchmark (both tilings): MFMA busy, LDS conflicts, wait buckets.  usage: r5_pmc_x3.sh <outdir>
out=${1:-gpurun_out/r5h}; mkdir -p $out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$out/counters.txt 2>&1
grep -o "SQ_LDS[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INST_CYCLES[A-Z_]*" $R/$out/counters.txt | sort -u | tr '\n' ' ' > $R/$out/counter_names.txt
cat $R/$out/counter_names.txt; echo
cd $R
for t in 128 256; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $out/pmc_${t}_$tag
    CAPMI_X3_TILE=$t timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_${t}_$tag -- python scripts/tools_x3w_bench.py --short > $out/pmc_${t}_$tag.log 2>&1
    tail -2 $out/pmc_${t}_$tag.log
  done
done
python - <<PY
import csv, glob, os, collections, json
res={}
for d in sorted(glob.glob('$out/pmc_*_*')):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d,'**','*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'gemm_x3' not in k: continue
            acc[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
    res[os.path.basename(d)]={k:{c:round(sum(v)/len(v)) for c,v in cs.items()} | {'launches': max(len(v) for v in cs.values())} for k,cs in acc.items()}
json.dump(res, open('$out/x3_pmc.json','w'), indent=1)
for d,ks in res.items():
    for k,cs in ks.items(): print(d, k[-44:], cs)
PY
find $out -name "*.csv" -size +5M -delete 2>/dev/null; du -sh $out | tail -1
