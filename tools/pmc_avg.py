"""Average PMC counters per dispatch of one kernel from a rocprofv3 --pmc run: python tools/pmc_avg.py <dir> <kernel substring>"""
import csv
import glob
import sys
from collections import defaultdict

d, pat = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
acc, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f)):
    if pat in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        cnt[r['Counter_Name']] += 1
for k in sorted(acc):
    print(f'{k:32s} {acc[k] / cnt[k]:16.1f}  ({cnt[k]} dispatches)')
