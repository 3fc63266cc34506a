cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
for v in 8 0; do
  rm -rf $R/prof_m$v
  UENC_MSDA_VARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_m$v -o p -- python3 bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $R/prof_m$v.log 2>&1 || exit 1
  f=$(find $R/prof_m$v -name 'p_kernel_stats.csv' | head -1)
  echo "== variant $v"; grep -E "msda_bwd_bin|msda_bin_reduce|msda_tiled" $f | cut -d, -f1-4
  rm -rf $R/prof_m$v
done
