set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_exact_gpu.py -x -q -m gpu -k "mha or decoder or model or attention or full" > $R/r3_t25.log 2>&1 || { tail -40 $R/r3_t25.log; exit 1; }
tail -3 $R/r3_t25.log
rm -rf $R/prof_k
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_k -o p -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $R/r3_prof_k.log 2>&1 || exit 1
python - <<PYEOF
import csv,glob
f=glob.glob("gpurun_out/prof_k/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("mha_",)): print("STAT", r["Name"][:50], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PYEOF
rm -rf $R/prof_k
timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['step_ms'])"
echo DONE
