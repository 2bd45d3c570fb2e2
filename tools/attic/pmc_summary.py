"""Average PMC counters per kernel from a rocprofv3 rocpd database. Usage: pmc_summary.py X_results.db [filter]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else 'k_'
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall() \
    if 'kernel_name' in cols else []
if not rows:
    print('columns:', cols)
for k, c, v, n in rows:
    if flt in k:
        print(f"{k[:60]:<60} {c:<28} {v:>18.1f} (n={n})")
