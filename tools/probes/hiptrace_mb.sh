#!/bin/bash
# HIP API calls recorded inside the capture of the mini-batch step (memcpy / memset nodes would show here, not in a kernel trace).
#   bash tools/probes/hiptrace_mb.sh [bench.py flags...]   -> gpurun_out/hiptrace_mb.txt
set -e -o pipefail
export TMPDIR=/tmp
rm -rf /tmp/ht_mb
rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/ht_mb -o t -- python3 bench.py --config mb --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-check "$@" > /dev/null 2> gpurun_out/hiptrace_mb.err
python3 - "$(find /tmp/ht_mb -name '*hip_api_trace.csv' | head -1)" > gpurun_out/hiptrace_mb.txt <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
incap, c, seq, last, last_seq = False, collections.Counter(), [], None, None
for r in rows:
    f = r['Function']
    if 'BeginCapture' in f:
        incap = True; c.clear(); seq = []; continue
    if 'EndCapture' in f:
        incap = False; last, last_seq = dict(c), list(seq); continue
    if incap:
        c[f] += 1
        if 'GetLastError' not in f and 'PeekAtLastError' not in f:
            seq.append(f)
print('calls inside the LAST capture:', {k: v for k, v in (last or {}).items() if 'GetLastError' not in k and 'PeekAtLastError' not in k})
run = []
for f in last_seq or []:
    if run and run[-1][0] == f:
        run[-1][1] += 1
    else:
        run.append([f, 1])
print(' '.join(f'{f.replace("hip", "")}x{n}' if n > 1 else f.replace('hip', '') for f, n in run))
PY
rm -rf /tmp/ht_mb
