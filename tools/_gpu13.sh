set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
# A. default bench line (all legs, incl. the CPU baseline)
timeout -k 10 900 python bench.py > $R/r03_bench_default.json 2> $R/r03_bench_default.err || exit 1
# B. kernel trace + stats of the bench command, on the same box
rm -rf $R/prof_k
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_k -o p -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $R/r3_prof_k.log 2>&1 || exit 1
f=$(find $R/prof_k -name 'p_kernel_trace.csv' | head -1)
PHASE_DETAIL=0 python tools/step_breakdown.py $f 90 > $R/r03_step_breakdown.txt || exit 1
cp $(find $R/prof_k -name 'p_kernel_stats.csv' | head -1) $R/r03_bench_kernel_stats.csv
grep '^{' $R/r3_prof_k.log | tail -1 > $R/r03_bench_under_rocprof.json
rm -rf $R/prof_k
# C. PMC passes
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/pmc_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/pmc_$c -o c -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/r3_pmc_$c.log 2>&1 || exit 1
done
python tools/pmc_traffic.py $R/pmc_FETCH_SIZE/c_counter_collection.csv $R/pmc_WRITE_SIZE/c_counter_collection.csv $R/r03_pmc_traffic.json > $R/r03_pmc_traffic.txt || exit 1
rm -rf $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/pmc_m
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/pmc_m -o m -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/r3_pmc_m.log 2>&1 || exit 1
python tools/mfma_util.py $R/pmc_m/m_counter_collection.csv $R/r03_mfma_util.json > /dev/null || exit 1
rm -rf $R/pmc_m
timeout -k 10 300 python tools/gemm_time_by_shape.py > $R/r03_gemm_shapes.txt 2>&1 || exit 1
# D. the bench line again with the PMC summaries of THESE sources in place (they are quoted only when the hash matches)
cp $R/r03_pmc_traffic.json $R/r03_mfma_util.json profiles/
timeout -k 10 400 python bench.py --steps 8 --no-cpu-baseline --no-extras > $R/r03_bench_quoting_pmc.json 2>/dev/null || exit 1
echo DONE
UENC_BENCH_BACKBONE=dinat timeout -k 10 400 python bench.py --steps 6 --warmup 3 --no-extras --no-cpu-baseline > $R/r03_bench_dinat.json 2>/dev/null || exit 1
echo DONE2
