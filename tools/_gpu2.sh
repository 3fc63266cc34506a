set -x
mkdir -p gpurun_out
timeout -k 10 600 python tools/diag_fullsize_error.py > gpurun_out/r3_diag_fullsize_error.txt 2>&1 && \
UENC_PARITY_OUT=gpurun_out/r03_parity_b.json timeout -k 10 600 python -m pytest tests/test_dinat_gpu.py -x -q -m gpu -k full_size_dinat > gpurun_out/r3_dinat_full.log 2>&1
echo rc=$?
tail -20 gpurun_out/r3_diag_fullsize_error.txt; tail -5 gpurun_out/r3_dinat_full.log
