"""Per-kernel and per-phase breakdown of the last full bench step in a rocprofv3 --kernel-trace CSV.
usage: python tools/step_breakdown.py gpurun_out/<dir>/p_kernel_trace.csv [top]
Steps are delimited by the one-per-step cast_multi_kernel launch; phases by the first launch of a phase's signature kernel."""
import collections, csv, os, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
NAMEW = int(os.environ.get("NAME_WIDTH", "64"))
st = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:NAMEW]) for r in rows)
idx = [i for i, r in enumerate(st) if 'cast_multi' in r[2]]
seg = st[idx[-2]:idx[-1]]
t0 = seg[0][0]
def first(name):
    for s, e, n in seg:
        if name in n: return (s - t0) / 1e6
    return None
wall = (seg[-1][1] - t0) / 1e6
busy = sum(e - s for s, e, n in seg) / 1e6
print(f"step wall {wall:.2f} ms  busy {busy:.2f} ms  launches {len(seg)}")
marks = [('fwd backbone', 0.0), ('fwd pixel decoder', first('msda_')), ('fwd transformer decoder + upsample', first('mha_q_kernel<0')),
         ('bwd transformer decoder', first('mha_dkdv')), ('bwd pixel decoder', first('msda_bwd_bin')), ('bwd backbone', first('wattn_bwd'))]
marks = [(n, t) for n, t in marks if t is not None] + [('end', 1e18)]
for (name, lo), (_, hi) in zip(marks, marks[1:]):
    b = sum(e - s for s, e, n in seg if lo <= (s - t0) / 1e6 < hi) / 1e6
    c = sum(1 for s, e, n in seg if lo <= (s - t0) / 1e6 < hi)
    print(f"  {name:36s} wall {min(hi, wall) - lo:7.2f} ms  busy {b:7.2f} ms  launches {c:5d}")
import os
if os.environ.get("PHASE_DETAIL"):
    for (name, lo), (_, hi) in zip(marks, marks[1:]):
        pt, pc = collections.Counter(), collections.Counter()
        for s, e, n in seg:
            if lo <= (s - t0) / 1e6 < hi:
                pt[n] += e - s; pc[n] += 1
        print(f"---- {name}")
        for n, t in pt.most_common(int(os.environ["PHASE_DETAIL"])):
            print(f"   {t / 1e6:8.3f} {pc[n]:5d}  {n}")
if os.environ.get("PHASE_SEQ"):        # the ordered launch sequence of the phases whose name contains the string: offset, duration, idle gap before it
    for (name, lo), (_, hi) in zip(marks, marks[1:]):
        if os.environ["PHASE_SEQ"] not in name: continue
        print(f"==== sequence of {name}")
        prev = None
        for s, e, n in seg:
            if lo <= (s - t0) / 1e6 < hi:
                print(f"  {(s - t0) / 1e3:10.1f} us  {(e - s) / 1e3:8.1f}  gap {((s - prev) / 1e3 if prev else 0):7.1f}  {n}")
                prev = e
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in seg:
    tot[n] += e - s; cnt[n] += 1
for n, t in tot.most_common(top):
    print(f"{t / 1e6:8.3f} {cnt[n]:5d}  {n}")
