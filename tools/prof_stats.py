"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats output directory (and optionally save it)."""
import csv
import glob
import sys

d = sys.argv[1]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None
for f in sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    hdr = f'{"kernel":90s} {"calls":>6s} {"avg_us":>10s} {"total_ms":>10s} {"pct":>6s}'
    lines = [hdr]
    for r in rows[:40]:
        lines.append(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.1f} {float(r["TotalDurationNs"]) / 1e6:10.2f} {float(r["Percentage"]):6.2f}')
    print('\n'.join(lines))
    if out:
        out.write('\n'.join(lines) + '\n')
