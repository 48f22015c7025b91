python -m pytest tests/test_kernels_gpu.py tests/test_full_size_parity_gpu.py tests/test_model_api_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/add_t.log
bash scripts/prof_config.sh mh_txe transformer_xe > /dev/null 2>&1
bash scripts/prof_config.sh mh_uxe updown_xe > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_txe gpurun_out/prof_mh_uxe
cat gpurun_out/add_t.log; head -14 gpurun_out/mh_txe_kernel_stats.md | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_mh_txe.log | head -1;  head -14 gpurun_out/mh_uxe_kernel_stats.md | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_mh_uxe.log | head -1
