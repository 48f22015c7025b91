#!/bin/bash
# A/B of research switches inside ONE gpurun call (boxes differ by up to 10 %): every line = one `bench.py --brief` of the variants
# build (scripts/build_variants.sh).   usage: r4_ab.sh "VAR=a VAR2=b" "VAR=c ..." ...   (each argument = one configuration)
export CAPMI_LIB=$PWD/variants/libcapmi.so
out=gpurun_out/r4_ab.log; : > $out
run() { echo "== $*" >> $out; env $* python bench.py --brief --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; a = d['attention']
        print('ms_per_step %.3f  gemm_lc avg %.2f us frac %.4f  attention avg %.2f us' % (d['ms_per_step'], r['avg_launch_us'], r['frac'], a['avg_launch_us']))" >> $out; }
for rep in 1 2; do for cfg in "$@"; do run $cfg; done; done
cat $out
