#!/bin/bash
# A/B of a Python engine file inside ONE gpurun call: usage r4_ab_file.sh <config> <path in repo> <old copy>
cfg=$1; f=$2; old=$3
cp $f /tmp/new_file.py
one() { python bench.py --config $cfg --brief --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do cp $old $f; echo "old $(one)"; cp /tmp/new_file.py $f; echo "new $(one)"; done
