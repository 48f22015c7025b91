python -m pytest tests/test_kernels_gpu.py tests/test_full_size_parity_gpu.py tests/test_model_api_gpu.py tests/test_updown_gpu.py -q -x 2>&1 | tail -3 > gpurun_out/add_t.log
bash scripts/prof_config.sh mh_txe transformer_xe > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_txe
cat gpurun_out/add_t.log; grep -E "colsum|dropout_mask|kernel time" gpurun_out/mh_txe_kernel_stats.md | cut -c1-150
