"""HBM traffic of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write "gconv_kernel<128, 64, 32, 32, 1>"

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores.
"""
import csv
import glob
import json
import sys


def mean_counter(d, kernel, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
            if r['Counter_Name'] == counter and kernel in r['Kernel_Name'].replace('(anonymous namespace)::', '')]
    return sum(vals) / max(len(vals), 1), len(vals)


fetch, nf = mean_counter(sys.argv[1], sys.argv[3], 'FETCH_SIZE')
write, nw = mean_counter(sys.argv[2], sys.argv[3], 'WRITE_SIZE')
out = {'kernel': sys.argv[3], 'launches_sampled': [nf, nw], 'fetch_size_kib_raw': fetch, 'write_size_kib': write,
       'hbm_read_bytes_per_launch': 2.0 * fetch * 1024, 'hbm_write_bytes_per_launch': write * 1024,
       'hbm_bytes_per_launch': (2.0 * fetch + write) * 1024,
       'note': 'FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B); separate --pmc passes'}
print(json.dumps(out, indent=1))
