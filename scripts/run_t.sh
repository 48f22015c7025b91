CAPMI_DW_STREAM=0 bash scripts/prof_config.sh mh_txe1 transformer_xe > /dev/null 2>&1
bash scripts/prof_config.sh mh_txe2 transformer_xe > /dev/null 2>&1
rm -rf gpurun_out/prof_mh_txe1 gpurun_out/prof_mh_txe2
head -4 gpurun_out/mh_txe1_kernel_stats.md | cut -c1-150; head -4 gpurun_out/mh_txe2_kernel_stats.md | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_mh_txe1.log | head -1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_mh_txe2.log | head -1
