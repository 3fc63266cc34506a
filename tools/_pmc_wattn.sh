cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
rm -rf $R/pmc_w
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/pmc_w -o a -- python3 tools/wattn_one.py > $R/pmc_w_a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/pmc_w -o b -- python3 tools/wattn_one.py > $R/pmc_w_b.log 2>&1 || exit 1
for f in a b; do c=$(find $R/pmc_w -name "${f}_counter_collection.csv" | head -1); echo "== $c"; python tools/pmc_rows.py $c wattn_bwd; done > $R/wattn_bwd_pmc_new.txt
cat $R/wattn_bwd_pmc_new.txt
