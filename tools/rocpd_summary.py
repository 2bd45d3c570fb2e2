"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel table: calls, total, mean, share.
Usage: python tools/rocpd_summary.py gpurun_out/prof/X_results.db [top_n] > profiles/rNN_*.txt"""
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {path}\n# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'share':>6}")
    for n, c, t, a, mn, mx in rows[:top]:
        n = n if len(n) <= 88 else n[:85] + '...'
        print(f"{n:<90} {c:>7} {t / 1e6:>10.3f} {a / 1e3:>10.2f} {mn / 1e3:>9.2f} {mx / 1e3:>9.2f} {100 * t / tot:>5.1f}%")


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)


def window(path, marker='k_compact', first=1, second=2, top=30):
    """Per-kernel summary restricted to the interval between two dispatches of `marker` (one bench step)."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    ts = [r[0] for r in cur.execute("select start from kernels where name like ? order by start", (f'%{marker}%',))]
    t0, t1 = ts[first], ts[second]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where start>=? and start<? "
                       "group by name order by 3 desc", (t0, t1)).fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# window between {marker} dispatch #{first} and #{second}: wall {(t1 - t0) / 1e6:.2f} ms, "
          f"kernel-busy {tot / 1e6:.2f} ms, {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':<100} {'calls':>6} {'total_ms':>9} {'avg_us':>10} {'share':>6}")
    for n, c, t, a in rows[:top]:
        n = n if len(n) <= 98 else n[:95] + '...'
        print(f"{n:<100} {c:>6} {t / 1e6:>9.3f} {a / 1e3:>10.2f} {100 * t / tot:>5.1f}%")
