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
print('\n## launches of k_conv<3, 2, 2, 4, true, 1, ...> (every instantiation: plain, fused FPN merge, seven-tile mode) by grid; the 196->196 '
      '@240x320 roofline launch of bench.py is grid x = 9830400 threads (64 images x 30 x 20 tiles x 256 threads)')
if gcols:
    g = gcols[0] if 'grid_size' not in cols else 'grid_size'
    sel = ', '.join(gcols[:3])
    # the template prefix, not one instantiation: round 3 added two trailing template arguments and the exact-name filter of
    # rounds 1-2 matched nothing (the table was empty in profiles/r03_*)
    for row in cur.execute(f"select {sel}, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                           f"where name like '%k_conv<3, 2, 2, 4, true, 1,%' group by {sel} order by 5 desc"):
        print('grid', row[:len(gcols[:3])], f'calls {row[-4]}  avg {row[-3]:.1f} us  min {row[-2]:.1f}  max {row[-1]:.1f}')
        if row[0] == 9830400:          # machine-readable: bench.py prints it next to its own event timing (roofline.launch_ms_rocprof)
            print(f'roofline_launch_rocprof: kernel=k_conv[K9 3x3 196->196 @240x320] calls={row[-4]} avg_us={row[-3]:.1f}')
    print('\n## launches of k_wino<...> (K17, Winograd F(2x2,3x3): the stride-1 3x3 layers of the inference step) by grid; the 196->196 @240x320 '
          'roofline launch of bench.py is grid x = 39321600 threads (64 images x 15 x 20 tiles x 4 channel blocks x 512 threads)')
    for row in cur.execute(f"select {sel}, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels "
                           f"where name like '%k_wino<%' group by {sel} order by 5 desc"):
        print('grid', row[:len(gcols[:3])], f'calls {row[-4]}  avg {row[-3]:.1f} us  min {row[-2]:.1f}  max {row[-1]:.1f}')
        if row[0] == 39321600:
            # bench.py times this launch in isolation (kernel_rooflines: one warm-up + three timed launches, the LAST four of this grid
            # in the process); the launches before them are the layer inside the warm-up / timed steps, on the model's activations
            last = [r[0] for r in cur.execute(f"select (end-start)/1e3 from kernels where name like '%k_wino<%' and {gcols[0]}=39321600 "
                                              "order by start desc limit 3")]
            inside = cur.execute(f"select avg(end-start)/1e3, count(*) from kernels where name like '%k_wino<%' and {gcols[0]}=39321600 "
                                 f"and start < (select min(start) from (select start from kernels where name like '%k_wino<%' and "
                                 f"{gcols[0]}=39321600 order by start desc limit 4))").fetchone()
            print(f'roofline_launch_rocprof: kernel=k_wino[K17 3x3 196->196 @240x320] calls={len(last)} avg_us={sum(last) / len(last):.1f}'
                  f'   (the three isolated launches bench.py times; the same layer inside the steps: {inside[1]} launches, avg {inside[0]:.1f} us)')
else:
    print('no grid columns in', cols)
