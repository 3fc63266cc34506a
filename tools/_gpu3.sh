set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_a
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a -o p -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3_prof_a.log 2>&1 || exit 1
f=$(find gpurun_out/prof_a -name 'p_kernel_trace.csv' | head -1)
PHASE_DETAIL=25 python tools/step_breakdown.py $f 90 > gpurun_out/r3_step_breakdown_a.txt
cp $(find gpurun_out/prof_a -name 'p_kernel_stats.csv' | head -1) gpurun_out/r3_kernel_stats_a.csv
rm -rf gpurun_out/prof_a
head -12 gpurun_out/r3_step_breakdown_a.txt
