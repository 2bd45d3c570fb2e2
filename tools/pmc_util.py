"""Per-kernel averages of the counters of one rocprofv3 --pmc pass (rocpd database) next to the dispatch durations:
    python tools/pmc_util.py results.db [name-filter ...]
Prints, per (kernel, grid): launches, average duration, and every counter's average value per launch."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
filt = sys.argv[2:]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print('# counters_collection columns:', cols)
has_time = 'start' in cols and 'end' in cols
q = ("select kernel_name, grid_size, counter_name, avg(value), count(*)" + (", avg(end - start)" if has_time else "") +
     " from counters_collection group by kernel_name, grid_size, counter_name")
rows = cur.execute(q).fetchall()
out = {}
for r in rows:
    k, g, c, v, n = r[:5]
    if filt and not any(f in k for f in filt):
        continue
    e = out.setdefault((k, g), {'n': n})
    e[c] = v
    if has_time:
        e['dur_us'] = r[5] / 1e3
for (k, g), e in sorted(out.items(), key=lambda kv: -kv[1].get('dur_us', 0)):
    name = k if len(k) < 70 else k[:67] + '...'
    print(f'{name} grid={g} n={e["n"]} ' + ' '.join(f'{c}={v:.6g}' for c, v in e.items() if c != 'n'))
