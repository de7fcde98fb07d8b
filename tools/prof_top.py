"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/prof_top.py <dir> [steps] [n]"""
import csv
import glob
import sys

d, steps, top = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 25
f = sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:top]:
    print(f"{r['Name'][:70]:70s} calls/step {float(r['Calls']) / steps:7.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us  ms/step "
          f"{float(r['TotalDurationNs']) / steps / 1e6:7.3f}")
print(f'all kernels {tot / steps / 1e6:.3f} ms/step; {sum(float(r["Calls"]) for r in rows) / steps:.0f} kernels/step')
