#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass (counter_collection.csv) to per-kernel L2 hit rates
(MI355X_MICROARCH.md: hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)).
    python profiles/summarize_l2.py <counter_collection.csv>"""
import csv
import re
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r'gv::(k_[a-z_0-9]+)(<[^>]*>)?', r['Kernel_Name'])
        if m:
            acc[(m.group(1) + (m.group(2) or '')).replace(' ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    print(f'{"kernel":44s} {"launches":>8s} {"TCC_HIT":>14s} {"TCC_MISS":>14s} {"L2 hit rate":>12s}')
    for k in sorted(acc):
        h, m = acc[k].get('TCC_HIT_sum', []), acc[k].get('TCC_MISS_sum', [])
        if not h or not m:
            continue
        h, m = h[len(h) // 2:], m[len(m) // 2:]            # later launches: steady state
        ha, ma = sum(h) / len(h), sum(m) / len(m)
        print(f'{k:44s} {len(acc[k]["TCC_HIT_sum"]):8d} {ha:14.0f} {ma:14.0f} {ha / max(ha + ma, 1):12.3f}')


if __name__ == '__main__':
    main(sys.argv[1])
