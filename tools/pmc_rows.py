"""Print per-kernel sums of a rocprofv3 counter_collection.csv (last dispatch of each kernel name)."""
import collections, csv, re, sys
per = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    d = per.setdefault((k, r["Dispatch_Id"]), {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
last = {}
for (k, i), d in per.items(): last[k] = d
for k, d in last.items():
    print(k)
    for n, v in sorted(d.items()): print(f"   {n:32s} {v:16.0f}")
