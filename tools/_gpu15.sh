set -x
cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > $R/r3_t15.log 2>&1 || { tail -30 $R/r3_t15.log; exit 1; }
tail -3 $R/r3_t15.log
for v in 8388608 0 8388608 0; do
  UENC_GEMM_VARIANT=$v timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', d['ms_per_step'], d['step_ms'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['families'].items()})" || exit 1
done
echo DONE
