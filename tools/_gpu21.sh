set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_exact_gpu.py -x -q -m gpu > $R/r3_t21.log 2>&1 || { tail -40 $R/r3_t21.log; exit 1; }
tail -3 $R/r3_t21.log
for v in 0 1 0 1; do
  UENC_MSDA_FUSED_FWD=$v timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fusedfwd $v', d['ms_per_step'], d['step_ms'])" || exit 1
done
echo DONE
