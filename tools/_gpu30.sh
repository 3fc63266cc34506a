cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_exact_gpu.py -x -q -m gpu -k "gemm_nt_ln or pixel_decoder or full or model or deform" > $R/r3_t30.log 2>&1 || { tail -40 $R/r3_t30.log; exit 1; }
tail -2 $R/r3_t30.log
for v in 0 1 0 1; do
  UENC_GEMM_LN=$v timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_ln $v', d['ms_per_step'], d['step_ms'])" || exit 1
done
