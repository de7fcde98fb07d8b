"""Per-hipGraph-replay kernel table from a rocprofv3 --kernel-trace run (developer tool).

    python3 tools/prof_replay.py <dir with *kernel_trace.csv> [--marker adam_tick_kernel] [--per-step 2] [--by-grid] [--seq]

The trace of `bench.py` holds set-up, warm-up, the timed hipGraph replays, the roofline leg and the other configurations; a
`--stats` summary averages over all of them.  Here the trace is cut at every `per_step`-th launch of the marker kernel (the
Adam tick closes an optimiser phase: two per GAN step), the MODAL launch count per cut identifies the replays of the timed
step, and only cuts of exactly that count and the modal set of kernels are averaged -- eager steps, the roofline leg and
every other workload drop out.  Prints per kernel: launches per step, average duration, ms per step; then the span of a
replay (first start to last end), the kernel-time sum and the "< 8 us" line.
"""
import argparse
import collections
import csv
import glob
import re


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)
    return n[:72]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--marker', default='adam_tick_kernel')
    ap.add_argument('--per-step', type=int, default=2)
    ap.add_argument('--by-grid', action='store_true', help='split a kernel by its grid size (separates layer shapes)')
    ap.add_argument('--seq', action='store_true', help='print the launch sequence of one replay with durations')
    ap.add_argument('--top', type=int, default=60)
    a = ap.parse_args()
    fs = glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True)
    assert fs, 'no kernel_trace.csv under ' + a.dir
    rows = []
    for f in fs:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    cuts, cur, nmark = [], [], 0
    for r in rows:
        cur.append(r)
        if a.marker in r['Kernel_Name']:
            nmark += 1
            if nmark % a.per_step == 0:
                cuts.append(cur)
                cur = []
    counts = collections.Counter(len(c) for c in cuts)
    # the timed replays: the most frequent launch count, then the most frequent name sequence of that count
    n_modal = counts.most_common(1)[0][0]
    cand = [c for c in cuts if len(c) == n_modal]
    # (compared as multisets: a step with two graph branches starts its kernels in a slightly different order every replay)
    seqs = collections.Counter(tuple(sorted(r['Kernel_Name'] for r in c)) for c in cand)
    seq_modal = seqs.most_common(1)[0][0]
    steps = [c for c in cand if tuple(sorted(r['Kernel_Name'] for r in c)) == seq_modal]
    # graph replays are back to back: drop cuts whose span is far above the median (eager steps of the same sequence)
    span = sorted(int(c[-1]['End_Timestamp']) - int(c[0]['Start_Timestamp']) for c in steps)
    med = span[len(span) // 2]
    steps = [c for c in steps if int(c[-1]['End_Timestamp']) - int(c[0]['Start_Timestamp']) < 1.15 * med]
    ns = len(steps)
    agg = collections.OrderedDict()
    small_n = small_t = 0.0
    tot = 0.0
    for c in steps:
        for r in c:
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            key = short(r['Kernel_Name'])
            if a.by_grid:
                key += ' grid=%d' % (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']))
            e = agg.setdefault(key, [0, 0.0])
            e[0] += 1
            e[1] += d
            tot += d
            if d < 8.0:
                small_n += 1
                small_t += d
    print('# %d replays of %d launches averaged (of %d cuts at every %d-th %s; launch counts seen: %s)' % (
        ns, n_modal, len(cuts), a.per_step, a.marker, dict(counts.most_common(4))))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print('%-84s calls/step %6.1f  avg %8.2f us  ms/step %7.3f' % (k, n / ns, t / n, t / ns / 1e3))
    spans = [(int(c[-1]['End_Timestamp']) - int(c[0]['Start_Timestamp'])) / 1e6 for c in steps]
    print('kernel-time sum %.3f ms/step in a span of %.3f ms (first start to last end, mean of %d replays); %d kernels/step' % (
        tot / ns / 1e3, sum(spans) / ns, ns, n_modal))
    print('launches shorter than 8 us: %.0f per step = %.3f ms/step' % (small_n / ns, small_t / ns / 1e3))
    if a.seq:
        c = steps[len(steps) // 2]
        t0 = int(c[0]['Start_Timestamp'])
        prev_end = t0
        for r in c:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            print('%9.1f us  +gap %5.1f  dur %8.2f  grid %6d x%-3d %s' % (
                (s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
                int(r['Grid_Size_Y']), short(r['Kernel_Name'])))
            prev_end = e


if __name__ == '__main__':
    main()
