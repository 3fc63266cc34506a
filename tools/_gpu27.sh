cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attn_mask or decoder or full_model" 2>&1 | tail -2 || exit 1
rm -rf $R/prof_k
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_k -o p -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $R/r3_prof_k.log 2>&1 || exit 1
python - <<PYEOF
import csv,glob
f=glob.glob("gpurun_out/prof_k/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("attn_mask",)): print("STAT", r["Name"][:50], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PYEOF
rm -rf $R/prof_k
