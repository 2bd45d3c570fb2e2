"""Largest idle gaps between consecutive kernels of one bench step in a rocprofv3 rocpd database.
Usage: python tools/rocpd_gaps.py X_results.db [top]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ts = [r[0] for r in cur.execute("select start from kernels where name like '%k_compact%' order by start")]
t0, t1 = ts[1], ts[2]
rows = cur.execute("select name, start, end from kernels where start>=? and start<? order by start", (t0, t1)).fetchall()
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps.append((s1 - e0, n0[:70], n1[:70], (s0 - t0) / 1e6))
gaps.sort(reverse=True)
print(f'total idle {sum(g[0] for g in gaps if g[0] > 0) / 1e6:.2f} ms over {len(gaps)} gaps')
for g, a, b, at in gaps[:top]:
    print(f'{g / 1e6:8.3f} ms at +{at:7.2f} ms  after {a}\n{"":24}before {b}')
