python -m pytest tests/test_kernels_gpu.py tests/test_full_size_parity_gpu.py tests/test_updown_gpu.py -q -x 2>&1 | tail -3 > gpurun_out/add_t.log
bash scripts/prof_config.sh mh_uxe updown_xe > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_uxe
cat gpurun_out/add_t.log; grep -E "dpatt|colsum|kernel time" gpurun_out/mh_uxe_kernel_stats.md | cut -c1-150
