set -x
UENC_PARITY_OUT=gpurun_out/r03_parity.json timeout -k 10 1000 python -m pytest tests -x -q -m gpu --deselect tests/test_model_gpu.py::test_rccl_call_path_on_one_gpu -k "absolute_position or sequence or postproc or data_eval or inference_on_dataset or checkpoint or predictor or kitti" > gpurun_out/r03_gpu_suite2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_gpu_suite2.log
tail -4 gpurun_out/r03_gpu_suite2.log
