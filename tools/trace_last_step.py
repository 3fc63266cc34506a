"""Per-kernel breakdown of the last bench step from a rocprofv3 kernel trace CSV + the bench JSON line."""
import collections, csv, glob, json, sys
d = sys.argv[1]
log = open(d.rstrip('/') + '.log').read().splitlines()
rec = json.loads([l for l in log if l.startswith('{"metric')][-1])
print('value', rec['value'], 'ms/step', rec['ms_per_step'], 'roofline', rec['roofline']['achieved'], rec['roofline']['frac'])
tr = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(tr)))
st = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
end = st[-1][1]
win = rec['ms_per_step'] * 1e6
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in st:
    if s >= end - win:
        agg[n][0] += e - s; agg[n][1] += 1
tot = sum(v[0] for v in agg.values())
print('kernel time in last step window: %.1f ms of %.1f' % (tot / 1e6, win / 1e6))
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{t/1e6:8.2f} ms {c:5d}  {n[:110]}")
