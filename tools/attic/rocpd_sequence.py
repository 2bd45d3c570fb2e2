"""Dispatch sequence of one bench step (between two k_compact dispatches) from a rocprofv3 rocpd database:
index, start offset, duration, gap to the previous kernel's end, short name.  For finding glue kernels and idle gaps.
Usage: python tools/rocpd_sequence.py X_results.db [min_us]"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ts = [r[0] for r in cur.execute("select start from kernels where name like '%k_compact%' order by start")]
t0, t1 = ts[1], ts[2]
prev_end = None
for i, (name, s, e) in enumerate(cur.execute("select name, start, end from kernels where start>=? and start<? order by start", (t0, t1))):
    short = re.sub(r'\(anonymous namespace\)::|at::native::|void ', '', name)[:110]
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    if (e - s) / 1e3 >= min_us or gap > 20:
        print(f'{i:4d} t={(s - t0) / 1e6:8.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  {short}')
    prev_end = e if prev_end is None else max(prev_end, e)
