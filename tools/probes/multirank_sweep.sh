#!/bin/bash
# Functional sweep of `bench.py --gpus N` configurations with the ranks SHARING one GPU (GV_DIST_BACKEND=gloo; at most 4 ranks: the
# GPU boxes allow 6 processes on a card): every scheme / config must run and end with a finite loss.  Not a timing.
run() { tag=$1; shift; GV_DIST_BACKEND=gloo timeout -k 10 280 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > /tmp/o_$tag.json 2>/tmp/e_$tag.log; rc=$?; python - "$tag" $rc <<'PY'
import json,sys
tag,rc=sys.argv[1],sys.argv[2]
try:
    t=open(f'/tmp/o_{tag}.json').read().strip().split('\n')[-1]
    d=json.loads(t); print(tag, 'rc', rc, 'loss', d['final_loss'], 'ms', round(d['ms_per_step'],2), d['config'].get('partition'), d['config'].get('launch','')[:40])
except Exception as e:
    print(tag, 'rc', rc, 'ERR', open(f'/tmp/e_{tag}.log').read()[-600:].replace('\n',' | '))
PY
}
run g3 --gpus 3
run c4g2 --config c4 --gpus 2
run c4g3row --config c4 --gpus 3 --partition row
run weak4 --gpus 4 --scaling weak
run f32flows2 --gpus 2 --n-flows 3
run f32flows2row --gpus 2 --n-flows 3 --partition row
run c3g4 --config c3 --gpus 4
run sharded_row_c3 --config c3 --gpus 2 --partition row --sharded-adam
run mb2 --config mb --gpus 2
