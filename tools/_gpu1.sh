set -x
mkdir -p gpurun_out
export UENC_PARITY_OUT=gpurun_out/r03_parity_a.json
timeout -k 10 900 python -m pytest tests/test_exact_gpu.py tests/test_dinat_gpu.py -x -q -m gpu > gpurun_out/r3_parity_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_parity_tests.log
tail -5 gpurun_out/r3_parity_tests.log
timeout -k 10 300 python tools/gemm_time_by_shape.py > gpurun_out/r3_gemm_shapes_base.txt 2>&1
timeout -k 10 400 python bench.py --steps 10 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_base.json 2> gpurun_out/r3_bench_base.err
tail -c 1500 gpurun_out/r3_bench_base.json
