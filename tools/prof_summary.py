"""Dev tool: summarise a rocprofv3 kernel_stats csv (ms per step)."""
import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
f = sorted(glob.glob(d + '/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    if r['Name'].startswith('void at::') or 'rocclr' in r['Name'] or 'at::native' in r['Name']: continue
    ms = float(r['TotalDurationNs']) / 1e6 / steps; tot += ms
    if ms > 0.05: print(f"{r['Name'][:86]:86s} calls/step={int(r['Calls'])/steps:6.1f} ms/step={ms:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print('total (own kernels) ms/step', round(tot, 3))
