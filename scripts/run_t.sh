export PYTHONPATH=$PWD
for rep in 1 2; do
echo "prev (one instance, 128 VGPRs + spills)" >> gpurun_out/mb.log
CAPMI_LIB=$PWD/variants/libcapmi_prev.so python scripts/mha_ablate.py 0 2>&1 | grep "^abl" >> gpurun_out/mb.log
echo "new (512-thread instance 160 VGPRs)" >> gpurun_out/mb.log
python scripts/mha_ablate.py 0 2>&1 | grep "^abl" >> gpurun_out/mb.log
done
cat gpurun_out/mb.log
