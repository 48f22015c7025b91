#!/bin/bash
# Research build of libcapmi: -DCAPMI_VARIANTS makes capmi::research() / capmi::ablate_env() read the environment (the product
# build compiles them to constants) and instantiates the profiling-ablation kernels (CAPMI_LC_ABLATE, CAPMI_ARES_ABLATE,
# CAPMI_GEMM_ABLATE, CAPMI_SEL_ABLATE).  Output: variants/libcapmi.so (git-ignored; travels with gpurun); use it with
#   CAPMI_LIB=$PWD/variants/libcapmi.so python scripts/gemm_ablate.py
set -e
cd "$(dirname "$0")/.."
mkdir -p variants/obj
SRC=imagecaptioning/pytorch_amd/csrc
pids=()
for f in $SRC/*.hip; do
  b=$(basename "$f" .hip)
  extra=""
  case "$b" in gemm_x3|gemm_x3w|gemm_lc|sampler) extra="-fno-slp-vectorize";; decode_opts) extra="-ffp-contract=off";; esac       # = build.py EXTRA_FLAGS
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DCAPMI_VARIANTS $extra -c "$f" -o variants/obj/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcapmi.so variants/obj/*.o
echo variants/libcapmi.so
