#!/bin/bash
# PMC passes of the bench command (separate runs per counter, kernel-trace only: MI355X_MICROARCH.md / gpurun rules), summaries
# into gpurun_out/<tag>_pmc_traffic.json and <tag>_pmc_mfma.json.     usage: pmc_passes.sh tag
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${tag}_$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- $CMD > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
rm -rf $R/gpurun_out/pmc_${tag}_mfma
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_mfma -- $CMD > $R/gpurun_out/pmc_${tag}_mfma.log 2>&1
cd $R
python scripts/tools_pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/${tag}_pmc_traffic.json
python scripts/tools_pmc_mfma.py gpurun_out/pmc_${tag}_mfma gpurun_out/${tag}_pmc_mfma.json | tail -3
# keep the merged-back payload small
find gpurun_out/pmc_${tag}_* -name "*.csv" -size +20M -delete 2>/dev/null
du -sh gpurun_out/pmc_${tag}_* | tail -5
