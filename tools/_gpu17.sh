cd $GRAFT_REPO_ROOT
R=gpurun_out
for v in 0 4 11 19 3; do UENC_MSDA_VARIANT=$v timeout -k 10 300 python tools/msda_tiled_bench.py 2>&1 | grep -E "init" | sed "s/^/variant $v: /" || exit 1; done
