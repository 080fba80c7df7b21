#!/bin/bash
# the mini-batch step with an environment knob off / on, alternating on one box:  KNOB=GV_INDEX_BATCH tools/probes/mb_ab.sh [bench flags]
mkdir -p gpurun_out/r5c
for rep in 1 2; do for v in ${VALS:-0 1}; do
  env ${KNOB:-GV_INDEX_BATCH}=$v timeout -k 10 300 python bench.py --config ${CFG:-mb} --no-cpu-baseline --no-check "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${CFG:-mb} ${KNOB:-GV_INDEX_BATCH}=$v', round(d['ms_per_step'],4), d.get('ms_per_step_repeats'), 'loss', d.get('final_loss'))" || exit 1
done; done
