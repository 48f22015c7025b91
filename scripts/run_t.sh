for rep in 1 2; do for v in 12 100000; do
echo "DW_BATCH=$v transformer_xe $(CAPMI_DW_BATCH=$v python bench.py --config transformer_xe --steps 12 --warmup 3 --no-cpu-baseline --no-prof --brief 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" >> gpurun_out/dws.log
done; done
python -m pytest tests/test_kernels_gpu.py -q -x -k deferred 2>&1 | tail -2 >> gpurun_out/dws.log
cat gpurun_out/dws.log
