#!/bin/bash
# One parametrised A/B runner for a single gpurun call (replaces the r4_ab*/r5_ab* one-offs).
#   scripts/ab.sh <outdir> [suite[:<pytest -k expr>]] [<config>[@ENV=V[,ENV=V..]][#steps] ...]
# `suite` runs the GPU tests first (optionally -k filtered); every other word is one bench.py run:
#   updown_scst            headline config, default switches
#   transformer_xe@CAPMI_GRAPH_STEP=0     a config with environment switches (comma separated)
#   aoa_nsc#30             30 timed steps instead of the default (30 for the headline, 12 otherwise)
# Each run prints `<label> ms_per_step loss step_ms`; the JSON lines land in <outdir>/<label>.json.
out=${1:-gpurun_out/ab}; shift; mkdir -p "$out"; cd "${GRAFT_REPO_ROOT:-/root/repo}"
ms() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d['ms_per_step'], d.get('loss'), d.get('step_ms'), flush=True)
except Exception as e:
    print(sys.argv[2], 'FAILED', type(e).__name__, e, flush=True)
PY
}
i=0
for w in "$@"; do
  case "$w" in
    suite*) k="${w#suite}"; k="${k#:}"
      if [ -n "$k" ]; then timeout 900 python -m pytest tests -m gpu -q -x -k "$k" > "$out/suite.log" 2>&1
      else timeout 900 python -m pytest tests -m gpu -q -x > "$out/suite.log" 2>&1; fi
      tail -6 "$out/suite.log";;
    *) i=$((i+1)); cfg="${w%%[@#]*}"; rest="${w#$cfg}"; steps=""; envs=""
      case "$rest" in *#*) steps="${rest##*#}"; rest="${rest%%#*}";; esac
      envs="${rest#@}"; envs="${envs//,/ }"
      if [ "$cfg" = updown_scst ]; then args="--steps ${steps:-30} --warmup 5 --no-other-configs --no-cpu-baseline"
      else args="--config $cfg --steps ${steps:-12} --warmup 3 --brief --no-cpu-baseline"; fi
      label="$i.$cfg${envs:+.$(echo $envs | tr ' =' '__')}"
      env $envs timeout 300 python bench.py $args > "$out/$label.json" 2> "$out/$label.err"
      ms "$out/$label.json" "$label";;
  esac
done
