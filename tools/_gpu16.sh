set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "msdeform" > $R/r3_t16.log 2>&1 || { tail -40 $R/r3_t16.log; exit 1; }
tail -3 $R/r3_t16.log
timeout -k 10 300 python tools/msda_tiled_bench.py > $R/r3_msda_tiled.txt 2>&1 || { tail -20 $R/r3_msda_tiled.txt; exit 1; }

UENC_MSDA_TILE=8,16,52 timeout -k 10 300 python tools/msda_tiled_bench.py >> $R/r3_msda_tiled.txt 2>&1 || exit 1
UENC_MSDA_TILE=16,32,150 timeout -k 10 300 python tools/msda_tiled_bench.py >> $R/r3_msda_tiled.txt 2>&1 || exit 1
cat $R/r3_msda_tiled.txt
echo DONE
