#!/bin/bash
# A/B of GV_MADE_CHAIN_PASSES (passes of a MADE's backward per gv_made_chain_iafb launch) over the bf16 flow configurations.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4p
run() { tag=$1; shift; for v in $PASSES; do GV_MADE_CHAIN_PASSES=$v timeout -k 10 200 python bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-check > gpurun_out/r4p/${tag}_p$v.json 2>gpurun_out/r4p/${tag}_p$v.err; python - "$tag" $v <<'PY'
import json,sys
tag,v=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open(f'gpurun_out/r4p/{tag}_p{v}.json').read().strip().split('\n')[-1]); print(tag,'passes',v,'ms',round(d['ms_per_step'],4))
except Exception as e: print(tag,v,'ERR',e)
PY
done; }
PASSES="${PASSES:-1 2 3 6}"
run c3 --config c3
run c2f3bf16 --n-flows 3 --gemm-precision bf16
run mbf3bf16 --config mb --n-flows 3 --gemm-precision bf16
