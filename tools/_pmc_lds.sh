# LDS bank-conflict hunt over one bench step: per kernel family, conflict cycles against LDS-active cycles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out
rm -rf $R/pmc_lds
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/pmc_lds -o l -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $R/pmc_lds.log 2>&1 || exit 1
python3 - <<'PY'
import csv, collections, re, glob
f = glob.glob('gpurun_out/pmc_lds/**/l_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0))
print(f"{'kernel':70s} {'launches':>8s} {'conflict':>12s} {'lds_active':>12s} {'ratio':>6s} {'busy':>12s}")
for k, d in rows[:30]:
    c, a = d.get("SQ_LDS_BANK_CONFLICT", 0), d.get("SQ_ACTIVE_INST_LDS", 0)
    print(f"{k:70s} {n[k]:8d} {c:12.0f} {a:12.0f} {c / max(a, 1):6.2f} {d.get('SQ_BUSY_CYCLES', 0):12.0f}")
PY
