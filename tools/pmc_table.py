"""Counters of rocprofv3 --pmc passes (one counter set per pass directory), averaged per dispatch and kernel, with the derived figures
the reviews ask for (developer tool):
    python3 tools/pmc_table.py <pass dir> [<pass dir> ...] [--kernels a,b,c] [--by-grid] [--top N]
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);  LDS conflict share = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE;  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md, HBM)."""
import argparse
import collections
import csv
import glob
import re


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', n)[:64]


ap = argparse.ArgumentParser()
ap.add_argument('dirs', nargs='+')
ap.add_argument('--kernels', default='')
ap.add_argument('--by-grid', action='store_true')
ap.add_argument('--top', type=int, default=12)
a = ap.parse_args()
want = [k for k in a.kernels.split(',') if k]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for d in a.dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if want and not any(w in k for w in want):
                continue
            if a.by_grid:
                k += ' grid=%d' % (int(r['Grid_Size']) // max(1, int(r['Workgroup_Size'])) if 'Grid_Size' in r else 0)
            e = acc[k][r['Counter_Name']]
            e[0] += float(r['Counter_Value'])
            e[1] += 1
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if want and not any(w in k for w in want):
                continue
            if a.by_grid:
                k += ' grid=%d' % (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']))
            e = dur[k]
            e[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            e[1] += 1
order = sorted(acc, key=lambda k: -dur[k][0])[:a.top]
for k in order:
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    n = max(v[1] for v in acc[k].values())
    line = '%-66s %4d dispatches/pass, %8.1f us under the counters' % (k, n, dur[k][0] / max(dur[k][1], 1))
    print(line)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        print('    MFMA-busy %5.1f %%  (SQ_VALU_MFMA_BUSY_CYCLES %.0f, GRBM_GUI_ACTIVE %.0f)' % (
            100.0 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0), c['SQ_VALU_MFMA_BUSY_CYCLES'], c['GRBM_GUI_ACTIVE']))
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
        print('    LDS bank conflicts %5.1f %% of LDS-active cycles  (%.0f / %.0f)' % (
            100.0 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], c['SQ_LDS_BANK_CONFLICT'], c['SQ_LDS_IDX_ACTIVE']))
    if 'SQ_WAIT_ANY' in c and c.get('SQ_WAVE_CYCLES'):
        print('    waves waiting %5.1f %% of wave cycles; VALU instructions %.0f, LDS instructions %.0f' % (
            100.0 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c.get('SQ_INSTS_VALU', float('nan')), c.get('SQ_INSTS_LDS', float('nan'))))
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        print('    HBM %.2f MB per launch (read 2 x %.0f KiB, written %.0f KiB)' % (
            (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6, c['FETCH_SIZE'], c['WRITE_SIZE']))
    rest = {n: v for n, v in c.items() if n not in ('FETCH_SIZE', 'WRITE_SIZE')}
    print('    ' + '  '.join('%s %.0f' % (n, v) for n, v in sorted(rest.items())))
