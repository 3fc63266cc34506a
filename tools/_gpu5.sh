set -x
export UENC_PARITY_OUT=gpurun_out/r03_parity_c.json
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_gpu_suite.log
tail -8 gpurun_out/r3_gpu_suite.log
