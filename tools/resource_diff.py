"""Diff two `hipcc -Rpass-analysis=kernel-resource-usage` logs (developer tool): kernels whose register counts moved.

    python tools/resource_diff.py OLD.txt NEW.txt
"""
import re
import subprocess
import sys


def parse(f):
    d, cur = {}, None
    for line in open(f):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            d[cur] = {}
        for key in ('VGPRs:', 'AGPRs', 'SGPRs:', 'Occupancy', 'ScratchSize', 'LDS Size'):
            if key in line and cur:
                m = re.search(key + r'[^\d]*(\d+)', line)
                if m:
                    d[cur][key] = int(m.group(1))
    return d


a, b = parse(sys.argv[1]), parse(sys.argv[2])
for k in b:
    if k in a and a[k] != b[k]:
        name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name)[:70]
        print(name, {x: (a[k][x], b[k][x]) for x in b[k] if a[k].get(x) != b[k][x]})
