cd $GRAFT_REPO_ROOT
R=gpurun_out
rm -f $R/r03_parity.json
UENC_PARITY_OUT=$R/r03_parity.json timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $R/r3_full_gpu.log 2>&1; rc=$?
tail -8 $R/r3_full_gpu.log
exit $rc
