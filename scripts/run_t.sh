export CAPMI_LIB=$PWD/variants/libcapmi.so PYTHONPATH=$PWD
python scripts/mha_ablate.py 0 2>&1 | grep "^abl" > gpurun_out/mha_ablate8.log
unset CAPMI_LIB
python -m pytest tests/test_full_size_parity_gpu.py tests/test_model_api_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -3 >> gpurun_out/mha_ablate8.log
bash scripts/prof_config.sh mh_txe transformer_xe > /dev/null 2>&1
bash scripts/prof_config.sh mh_aoa aoa_nsc > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_txe gpurun_out/prof_mh_aoa
cat gpurun_out/mha_ablate8.log; head -22 gpurun_out/mh_txe_kernel_stats.md | cut -c1-150;  head -24 gpurun_out/mh_aoa_kernel_stats.md | cut -c1-150
