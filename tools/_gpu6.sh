set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_w
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_w -o a -- python3 tools/wattn_one.py > gpurun_out/r3_pmc_w.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_w -o b -- python3 tools/wattn_one.py >> gpurun_out/r3_pmc_w.log 2>&1 || exit 1
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_w/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if "wattn_bwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in agg.items():
        print(f"  {k:28s} per launch {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
rm -rf gpurun_out/pmc_w
