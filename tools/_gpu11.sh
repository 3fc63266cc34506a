set -x
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_dinat_gpu.py -x -q -m gpu --deselect tests/test_model_gpu.py::test_rccl_call_path_on_one_gpu > gpurun_out/r3_defer_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3_defer_tests.log; tail -4 gpurun_out/r3_defer_tests.log
grep -q "rc=0" gpurun_out/r3_defer_tests.log || exit 1
for v in 0 1 0 1; do
  UENC_DEFER_SMALL=$v timeout -k 10 300 python bench.py --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEFER_SMALL=$v', d['ms_per_step'], d['step_ms'])" >> gpurun_out/r3_defer_ab.txt
done
cat gpurun_out/r3_defer_ab.txt
