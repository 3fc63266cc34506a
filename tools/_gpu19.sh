cd $GRAFT_REPO_ROOT
for v in 0 1 2; do UENC_MSDA_VARIANT=$v timeout -k 10 300 python tools/msda_fused_bench.py 2>&1 | grep -E "backward" | sed "s/^/variant $v: /" || exit 1; done
