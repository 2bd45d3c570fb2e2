"""Profile summary used for profiles/rNN_bench_*: whole-process table, one steady-state step, and the launches of the
roofline kernel by grid size (so that its rocprofv3 duration can be compared with bench.py's event timing)."""
import sqlite3, sys
sys.path.insert(0, 'tools')
import rocpd_summary as r
r.main(sys.argv[1], 12)
print()
r.window(sys.argv[1], top=45)
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
gcols = [c for c in cols if 'grid' in c.lower()]
print('\n## launches of k_conv<3, 2, 2, 4, true, 1> by grid (the 196->196 @240x320 roofline launch is grid x = 9830400 threads)')
if gcols:
    g = gcols[0] if 'grid_size' not in cols else 'grid_size'
    sel = ', '.join(gcols[:3])
    for row in cur.execute(f"select {sel}, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                           f"where name like '%k_conv<3, 2, 2, 4, true, 1, false>%' group by {sel} order by 5 desc"):
        print('grid', row[:len(gcols[:3])], f'calls {row[-4]}  avg {row[-3]:.1f} us  min {row[-2]:.1f}  max {row[-1]:.1f}')
else:
    print('no grid columns in', cols)
