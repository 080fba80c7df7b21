#!/bin/bash
# The dispatches of the LAST replayed step of a bench configuration, in start order: offset from the step's first dispatch, duration,
# queue, kernel.   bash tools/probes/timeline.sh <tag> [bench.py flags...]   -> gpurun_out/timeline_<tag>.txt
set -e -o pipefail
tag=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --no-check "$@" > /dev/null 2> gpurun_out/timeline_$tag.err
python3 - "$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)" > gpurun_out/timeline_$tag.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name'] or 'k_clip_adam' in r['Kernel_Name']]
lo, hi = marks[-2] + 1, marks[-1] + 1
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
PY
rm -rf /tmp/tl_$tag
