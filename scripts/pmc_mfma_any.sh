#!/bin/bash
# MFMA-busy PMC pass of any python command: pmc_mfma_any.sh tag script [args...] -> gpurun_out/<tag>_pmc_mfma.json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_${tag}_mfma
cd $R
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_mfma -- python "$@" > $R/gpurun_out/pmc_${tag}_mfma.log 2>&1
python scripts/tools_pmc_mfma.py gpurun_out/pmc_${tag}_mfma gpurun_out/${tag}_pmc_mfma.json | tail -12
find gpurun_out/pmc_${tag}_* -name "*.csv" -size +20M -delete 2>/dev/null
