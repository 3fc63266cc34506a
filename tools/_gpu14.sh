set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
rm -rf $R/prof_k
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_k -o p -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $R/r3_prof_k.log 2>&1 || exit 1
f=$(find $R/prof_k -name 'p_kernel_trace.csv' | head -1)
NAME_WIDTH=150 PHASE_SEQ="transformer decoder" python tools/step_breakdown.py $f 10 > $R/r3_phase_seq.txt || exit 1
grep '^{' $R/r3_prof_k.log | tail -1 > $R/r3_bench_under_rocprof_mid.json
rm -rf $R/prof_k
echo DONE
