"""Mean of each counter per kernel from a rocprofv3 --pmc counter_collection.csv:  python3 tools/pmc_kernel.py <csv> [name filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    acc[name[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for kname, cs in acc.items():
    print(kname, ' '.join(f'{c}={sum(v) / len(v):.4g}' for c, v in sorted(cs.items())), f'n={len(next(iter(cs.values())))}')
