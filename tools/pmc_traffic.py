"""HBM-side traffic per kernel family from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE, separate runs):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/r02_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md "HBM" prescribes: the counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced streaming reads (doubled here); WRITE_SIZE is exact for 16-byte streaming stores and float atomics.
Infinity-cache hits are counted (memory-side of L2), so this is L2 <-> fabric traffic, an upper bound of HBM traffic."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mfma_util import git_head, sources_sha

def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[re.sub(r'\(.*', '', r["Kernel_Name"]).replace("void ", "")][0] += float(r["Counter_Value"]) * 1024.0
            agg[re.sub(r'\(.*', '', r["Kernel_Name"]).replace("void ", "")][1] += 1
    return agg

fe, wr = load(sys.argv[1]), load(sys.argv[2])
fam = collections.defaultdict(lambda: {"launches": 0, "fetch_bytes": 0.0, "write_bytes": 0.0})
for k, (b, n) in fe.items():
    f = re.sub(r'<.*', '', k)
    fam[f]["launches"] += n; fam[f]["fetch_bytes"] += 2.0 * b
for k, (b, n) in wr.items():
    fam[re.sub(r'<.*', '', k)]["write_bytes"] += b
out = {}
for f, d in sorted(fam.items(), key=lambda x: -(x[1]["fetch_bytes"] + x[1]["write_bytes"]))[:16]:
    d["traffic_bytes_per_launch"] = (d["fetch_bytes"] + d["write_bytes"]) / max(d["launches"], 1)
    out[f] = {k: (round(v) if isinstance(v, float) else v) for k, v in d.items()}
    print(f"{f[:44]:44s} launches {d['launches']:5d}  fetch {d['fetch_bytes']/1e9:8.2f} GB  write {d['write_bytes']/1e9:8.2f} GB  per launch {d['traffic_bytes_per_launch']/1e6:8.1f} MB")
if len(sys.argv) > 3:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1; FETCH_SIZE x2 (gfx950)",
               "git_head": git_head(), "kernel_sources_sha": sources_sha(),       # bench.py quotes this file only while the kernel sources are unchanged
               "families": out}, open(sys.argv[3], "w"), indent=1)
