#!/bin/bash
# A/B of GV_MADE_FWD_PASSES (passes of a MADE's forward per gv_made_chain_fwd launch) over the bf16 flow configurations, alternating.
cd "$(dirname "$0")/../.."
one() { tag=$1; v=$2; shift 2; GV_MADE_FWD_PASSES=$v timeout -k 10 200 python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag fwd_passes', $v, 'ms', round(d['ms_per_step'],4), flush=True)"; }
for i in 1 2; do for v in ${PASSES:-1 6}; do one c3 $v --config c3; done; done
for i in 1 2; do for v in ${PASSES:-1 6}; do one c2f3bf16 $v --n-flows 3 --gemm-precision bf16; done; done
for v in ${PASSES:-1 6}; do one mbf3bf16 $v --config mb --n-flows 3 --gemm-precision bf16; done
