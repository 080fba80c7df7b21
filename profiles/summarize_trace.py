#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel calls/avg/total for the LAST n launches-per-step
window (the hipGraph replays of the timed region) plus the whole-run stats.
    python profiles/summarize_trace.py <kernel_trace.csv> [steps_in_window]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = name.replace('void ', '').replace('at::native::', '')
    return name[:110]


def main(path, steps=10):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # the timed region = the last `steps` occurrences of a once-per-step kernel
    marker = [i for i, r in enumerate(rows) if 'k_distmult_bce' in r['Kernel_Name']]
    lo = marker[-steps] if len(marker) >= steps else 0
    # walk back to the start of that step: first kernel after the previous step's last Adam kernel is fine to approximate
    win = rows[lo:]
    t_first, t_last = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
    agg = defaultdict(lambda: [0, 0])
    for r in win:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        a = agg[short(r['Kernel_Name'])]
        a[0] += 1
        a[1] += d
    busy = sum(v[1] for v in agg.values())
    print(f'window: {len(win)} launches over {(t_last - t_first) / 1e3 / steps:.1f} us/step wall, '
          f'{busy / 1e3 / steps:.1f} us/step kernel-busy, {len(win) / steps:.0f} launches/step')
    # how much of the window has 0 / 1 / 2 / >= 3 kernels in flight (streams side by side: parallel branches of the replayed graph)
    ev = sorted([(int(r['Start_Timestamp']), 1) for r in win] + [(int(r['End_Timestamp']), -1) for r in win])
    depth, last, hist = 0, ev[0][0], [0, 0, 0, 0]
    for t, dlt in ev:
        hist[min(depth, 3)] += t - last
        last, depth = t, depth + dlt
    tot = max(1, sum(hist))
    print('kernels in flight: ' + ', '.join(f'{lbl} {100 * h / tot:.1f} %' for lbl, h in zip(('none', 'one', 'two', 'three or more'), hist)))
    print(f'{"kernel":112s} {"n/step":>7s} {"avg_us":>9s} {"us/step":>9s} {"%busy":>6s}')
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k:112s} {n / steps:7.1f} {tot / n / 1e3:9.2f} {tot / 1e3 / steps:9.1f} {100 * tot / busy:6.1f}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
