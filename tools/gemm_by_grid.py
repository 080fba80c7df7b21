#!/usr/bin/env python3
"""Break one kernel's launches in a rocprofv3 kernel trace down by grid shape (last `steps` steps of the run).
    python tools/gemm_by_grid.py <kernel_trace.csv> <kernel substring> [steps]"""
import csv
import sys
from collections import defaultdict


def main(path, sub, steps=10):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marker = [i for i, r in enumerate(rows) if 'k_distmult_bce' in r['Kernel_Name']]
    win = rows[marker[-steps]:] if len(marker) >= steps else rows
    agg = defaultdict(lambda: [0, 0])
    for r in win:
        if sub not in r['Kernel_Name']:
            continue
        key = (r['Kernel_Name'].split('(')[0][-40:], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
               int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
        agg[key][0] += 1
        agg[key][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    print(f'{"kernel":42s} {"grid":>16s} {"n/step":>7s} {"avg_us":>8s} {"us/step":>8s}')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k[0]:42s} {str(k[1:]):>16s} {n / steps:7.1f} {t / n / 1e3:8.2f} {t / 1e3 / steps:8.1f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)
