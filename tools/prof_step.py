"""Per-kernel breakdown of one bench step from a rocprofv3 rocpd database (kernels view).

usage: python tools/prof_step.py gpurun_out/profN/xxx_results.db [top]
Steps are delimited by the one-per-step cast_multi_kernel launch; the second-to-last full step is reported."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
rows = db.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'cast_multi' in r[0]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
tot, cnt = {}, {}
for n, s, e in seg:
    n = re.sub(r'\(.*', '', n)[:90]
    tot[n] = tot.get(n, 0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
busy = sum(tot.values())
print(f"step wall {(seg[-1][2]-seg[0][1])/1e6:.2f} ms  busy {busy/1e6:.2f} ms  launches {len(seg)}")
for n, t in sorted(tot.items(), key=lambda x: -x[1])[:top]:
    print(f"{t/1e6:8.3f} {cnt[n]:5d}  {n}")
