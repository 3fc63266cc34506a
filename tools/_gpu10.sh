set -x
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "msdeform or deform" > gpurun_out/r3_msda_fused_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3_msda_fused_tests.log; tail -5 gpurun_out/r3_msda_fused_tests.log
timeout -k 10 300 python tools/msda_fused_bench.py > gpurun_out/r3_msda_fused_bench.txt 2>&1; tail -3 gpurun_out/r3_msda_fused_bench.txt
