#!/bin/bash
# GV_MADE_CHAIN_PASSES x GV_GRADW_SPLIT_MAX_SIDE x GV_MADE_ROW_BLOCKS on c3 and on c2 + 3 IAF blocks (bf16): ms per step.
#   PL="1 6" SP="64 96 128 160" RB="1 2" tools/probes/passes_grid.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4p
one() { tag=$1; shift; timeout -k 10 200 python bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-check > gpurun_out/r4p/g.json 2>gpurun_out/r4p/g.err; python - "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r4p/g.json').read().strip().split('\n')[-1]); print(sys.argv[1],'ms',round(d['ms_per_step'],4), flush=True)
except Exception as e: print(sys.argv[1],'ERR',e)
PY
}
for pl in ${PL:-1 6}; do for sp in ${SP:-32 64 96 160}; do for rb in ${RB:-1 2 3}; do
  GV_MADE_CHAIN_PASSES=$pl GV_GRADW_SPLIT_MAX_SIDE=$sp GV_MADE_ROW_BLOCKS=$rb one "c3 passes=$pl split=$sp blocks=$rb" --config c3
done; done; done
for pl in ${PL:-1 6}; do for sp in ${SP2:-24 48 96}; do
  GV_MADE_CHAIN_PASSES=$pl GV_GRADW_SPLIT_MAX_SIDE=$sp one "c2f3 passes=$pl split=$sp" --n-flows 3 --gemm-precision bf16
done; done
