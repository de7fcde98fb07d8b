"""profiles/r06_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic_r6.sh: HBM bytes per launch of the Winograd
kernel on the layer shapes of the headline step, each keyed as bench.py names it ("wino_kernel<BN> MxNxK=...").

    python3 tools/pmc_traffic_r6.py <outdir> <workload>... > profiles/r06_traffic.json
"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
SHAPES = {'vgg256': (32, 24, 24, 256, 256), 'vgg256h': (16, 24, 24, 256, 256), 'vgg512': (32, 12, 12, 512, 512),
          'vgg512h': (16, 12, 12, 512, 512), 'vgg128': (32, 48, 48, 64, 128), 's18432x256x1152': (32, 24, 24, 128, 256),
          's73728x128x1152': (32, 48, 48, 128, 128), 's4608x256x4608': (32, 12, 12, 512, 256), 's2304x512x2304': (16, 12, 12, 256, 512)}


def mean_counter(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and 'wino_kernel<' in r['Kernel_Name']]
    vals = [float(r['Counter_Value']) for r in rows]
    name = re.search(r'wino_kernel<\d+>', rows[0]['Kernel_Name']).group(0) if rows else None
    return sum(vals) / max(len(vals), 1), len(vals), name


out = {'_how': 'round 6: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_traffic_r6.sh) of tools/pmc_workloads.py: '
               'the layer itself (3x3 / stride 1 / pad 1 + bias + ReLU at the step\'s batch and size) through the same entry point, '
               'kernel and plan the step uses, 20 launches each; FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B), WRITE_SIZE '
               'exact (MI355X_MICROARCH.md, HBM).  algorithmic bytes = input + output + 3x3 weights, fp32 (the kernel reads the '
               'Winograd-domain weights, 16/9 of that)'}
root = sys.argv[1]
for wl in sys.argv[2:]:
    n, h, w, cin, cout = SHAPES[wl]
    fetch, nf, name = mean_counter(f'{root}/{wl}.fetch', 'FETCH_SIZE')
    write, nw, _ = mean_counter(f'{root}/{wl}.write', 'WRITE_SIZE')
    if not name:
        continue
    alg = 4 * (n * h * w * (cin + cout) + 9 * cin * cout)
    tot = (2.0 * fetch + write) * 1024
    out[f'{name} MxNxK={n * h * w}x{cout}x{9 * cin}'] = {
        'hbm_bytes_per_launch': tot, 'hbm_read_bytes_per_launch': 2.0 * fetch * 1024, 'hbm_write_bytes_per_launch': write * 1024,
        'algorithmic_bytes_per_launch': alg, 'ratio': round(tot / alg, 2), 'workload': wl, 'launches_sampled': [nf, nw]}
print(json.dumps(out, indent=1))
