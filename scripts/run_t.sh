for o in 0 4; do echo "== CAPMI_ARES_OPT=$o"; CAPMI_ARES_OPT=$o python scripts/gemm_ablate.py 2>&1 | tail -1; CAPMI_ARES_OPT=$o CAPMI_ARES_ABLATE=16 python scripts/gemm_trace.py 2>&1 | grep -A20 "kh=0" ; done
CAPMI_ARES_OPT=4 python -m pytest tests/test_kernels_gpu.py tests/test_updown_gpu.py -q -x 2>&1 | tail -3
