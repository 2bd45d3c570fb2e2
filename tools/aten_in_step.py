"""Vendor / ATen kernels of one bench step (between two k_compact dispatches) with their full names:
python tools/aten_in_step.py <rocprofv3 results.db>"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
ts = [r[0] for r in cur.execute("select start from kernels where name like '%k_compact%' order by start")]
rows = cur.execute("select name, count(*), sum(end-start) from kernels where start>=? and start<? group by name order by 3 desc",
                   (ts[1], ts[2])).fetchall()
tot = 0.0
for n, c, t in rows:
    if 'anonymous namespace)::k_' in n and 'at::native' not in n or n.startswith('_ZN12_GLOBAL') or n.startswith('far_') or '_ZN7far' in n:
        continue
    tot += t
    short = re.sub(r'\s+', ' ', n)
    m = re.search(r'(\w+Functor\w*|\w+_kernel_cuda\w*|launch_\w+|\w+Op\b)', short)
    print(f'{t / 1e6:7.3f} ms {c:5d}  {short[:60]} ... {m.group(1) if m else ""} ... {short[-90:]}')
print(f'total {tot / 1e6:.2f} ms')
