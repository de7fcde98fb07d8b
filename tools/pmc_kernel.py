"""Developer tool: average PMC counters per kernel from rocprofv3 --pmc output directories (rocpd .db or csv).

    python tools/pmc_kernel.py <kernel-name-substring> <dir> [<dir> ...]
"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    pat = sys.argv[1]
    acc = defaultdict(lambda: [0.0, 0])
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if pat in r['Kernel_Name']:
                    a = acc[(r['Kernel_Name'][:50], r['Counter_Name'])]
                    a[0] += float(r['Counter_Value'])
                    a[1] += 1
        for f in glob.glob(os.path.join(d, '**', '*.db'), recursive=True):
            db = sqlite3.connect(f)
            cols = [c[1] for c in db.execute("pragma table_info('counters_collection')")]
            kn = 'kernel_name' if 'kernel_name' in cols else 'name'
            for name, counter, value in db.execute(f'select {kn}, counter_name, value from counters_collection'):
                if pat in name:
                    a = acc[(name[:50], counter)]
                    a[0] += float(value)
                    a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print(f'{k:50s} {c:32s} {v / n:16.1f}  (n={n})')


if __name__ == '__main__':
    main()
