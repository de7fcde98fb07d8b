"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel time per step.

    python tools/prof_summary.py gpurun_out/prof3 STEPS [TOPN]
"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
topn = int(sys.argv[3]) if len(sys.argv) > 3 else 32
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'spin_kernel' not in r['Name']]
tot = 0.0
for r in rows[:topn]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:58]
    t = float(r['TotalDurationNs']) / 1e6 / steps
    tot += t
    print('%-60s calls/step %7.1f  avg %8.1f us  ms/step %7.3f' % (n, int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, t))
print('sum of shown %.3f ms/step; all kernels %.3f ms/step; %.0f kernels/step' % (
    tot, sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps, sum(int(r['Calls']) for r in rows) / steps))
