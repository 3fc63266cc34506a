set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msdeform" > $R/r3_t20.log 2>&1 || { tail -40 $R/r3_t20.log; exit 1; }
tail -3 $R/r3_t20.log
timeout -k 10 300 python tools/msda_fused_bench.py 2>&1 | tail -3
UENC_MSDA_TILE_BWD=8,32,78 timeout -k 10 300 python tools/msda_fused_bench.py 2>&1 | tail -1
UENC_MSDA_TILE_BWD=8,16,52 timeout -k 10 300 python tools/msda_fused_bench.py 2>&1 | tail -1
echo DONE
