"""Every dispatch of one steady-state bench step in launch order (rocprofv3 --kernel-trace database): start offset, duration, grid,
kernel.  Usage: python tools/step_launches.py X_results.db [min_us]"""
import sqlite3
import sys

path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
cur = sqlite3.connect(path).cursor()
ts = [r[0] for r in cur.execute("select start from kernels where name like '%k_compact%' order by start")]
t0, t1 = ts[1], ts[2]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
sel = f"select name, start, end, {gx} from kernels" if gx else "select name, start, end, 0 from kernels"
rows = cur.execute(sel + " where start>=? and start<? order by start", (t0, t1)).fetchall()
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')[:100]
print(f'# step window {(t1 - t0) / 1e6:.2f} ms, {len(rows)} dispatches; columns: start ms, duration us, grid x, kernel')
for n, s, e, g in rows:
    if (e - s) / 1e3 >= min_us:
        print(f'{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:9.1f} {g:10d}  {short(n)}')
