#!/bin/bash
# Functional sweep of bench.py flag combinations on one GPU (3 steps each): rc 0, a finite loss and the in-line parity leg where it runs.
cd "$(dirname "$0")/../.."
run() { tag=$1; shift; timeout -k 10 280 python bench.py "$@" --steps 3 --warmup 2 --cpu-seconds 3 > /tmp/o_$tag.json 2>/tmp/e_$tag.log; rc=$?; python - "$tag" $rc <<'PY'
import json,sys
tag,rc=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(f'/tmp/o_{tag}.json').read().strip().split('\n')[-1])
    pc=d.get('parity_check') or {}
    print(tag, 'rc', rc, 'loss', round(d['final_loss'],4), 'ms', round(d['ms_per_step'],3), 'parity', pc.get('parity_max_rel_err'))
except Exception as e:
    print(tag, 'rc', rc, 'ERR', open(f'/tmp/e_{tag}.log').read()[-500:].replace('\n',' | '))
PY
}
run c2f3 --n-flows 3
run c2bf16 --gemm-precision bf16
run c2f1bf16 --n-flows 1 --gemm-precision bf16
run c4f3 --config c4 --n-flows 3
run c4f3bf16 --config c4 --n-flows 3 --gemm-precision bf16
run h100 --hidden 100
run nograph --no-graph
run mbf3 --config mb --n-flows 3
run c3f32 --config c3 --gemm-precision f32
