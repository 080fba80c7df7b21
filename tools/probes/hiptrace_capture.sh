cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for f in 3; do
  rocprofv3 --hip-runtime-trace --output-format csv -d gpurun_out/hiptrace_f$f -- python3 tools/probes/graph_nodes_probe.py $f > gpurun_out/hiptrace_f$f.log 2>&1
  python3 - <<PY
import csv, glob, collections
for fn in glob.glob('gpurun_out/hiptrace_f$f/**/*hip_api_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    incap=False; c=collections.Counter(); last=None
    for r in rows:
        fnm=r['Function']
        if 'BeginCapture' in fnm: incap=True; c.clear(); continue
        if 'EndCapture' in fnm: incap=False; last=dict(c); continue
        if incap: c[fnm]+=1
    print('flows $f: calls inside the LAST capture:', {k:v for k,v in (last or {}).items() if 'Launch' not in k and 'GetLastError' not in k and 'PeekAtLastError' not in k}, 'launches', sum(v for k,v in (last or {}).items() if 'Launch' in k))
PY
done
