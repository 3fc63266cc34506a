set -x
timeout -k 10 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || exit 1
tail -c 400 gpurun_out/r03_bench_default.json
rm -f gpurun_out/r03_parity.json
UENC_PARITY_OUT=gpurun_out/r03_parity.json timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gpu_suite.log
tail -4 gpurun_out/r03_gpu_suite.log
