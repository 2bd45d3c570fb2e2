"""Idle gaps between consecutive dispatches inside one bench step (rocprofv3 --kernel-trace database): where the GPU waits for
the host.  Usage: python tools/step_gaps.py X_results.db [min_gap_us]"""
import sqlite3
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
cur = sqlite3.connect(path).cursor()
ts = [r[0] for r in cur.execute("select start from kernels where name like '%k_compact%' order by start")]
t0, t1 = ts[1], ts[2]
rows = cur.execute("select name, start, end from kernels where start>=? and start<? order by start", (t0, t1)).fetchall()
busy = sum(e - s for _, s, e in rows)
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    gaps.append((g, n0, n1, (s1 - t0) / 1e6))
tot = sum(g for g, *_ in gaps if g > 0)
print(f'# step window {(t1 - t0) / 1e6:.2f} ms, kernel-busy {busy / 1e6:.2f} ms, {len(rows)} dispatches; idle between dispatches {tot / 1e3:.2f} ms '
      f'({sum(1 for g, *_ in gaps if g > min_gap)} gaps > {min_gap:.0f} us hold {sum(g for g, *_ in gaps if g > min_gap) / 1e3:.2f} ms)')
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:60]
for g, n0, n1, at in sorted(gaps, reverse=True)[:40]:
    if g > min_gap:
        print(f'{g:9.1f} us at {at:7.2f} ms   after {short(n0):<60}  before {short(n1)}')
