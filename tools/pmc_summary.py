#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of every counter over its dispatches.
    python tools/pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get('Kernel_Name', '')
                if flt and flt not in name:
                    continue
                acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    for name, ctrs in sorted(acc.items()):
        print(name[:110])
        for c, v in sorted(ctrs.items()):
            print(f'    {c:28s} n={len(v):4d}  mean={sum(v) / len(v):16.1f}')


if __name__ == '__main__':
    main()
