#!/bin/bash
# serialised + in-step kernel traces of ONE configuration (no PMC passes): tools/probes/trace_step.sh <config> <out-suffix> [bench flags...]
set -e -o pipefail
c=$1; sfx=$2; shift; shift
out=$PWD/gpurun_out/r5c
mkdir -p "$out"
export TMPDIR=/tmp
common="--no-cpu-baseline --no-check --profile-steps 0 --steps 10 --warmup 3"
for mode in instep serial; do
  rm -rf /tmp/pp
  if [ $mode = serial ]; then export GV_BWD_SIDE=0 GV_RGCN_BWD_SIDE=0 GV_MADE_PREPARE=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 bench.py --config $c $common "$@" > /dev/null 2> "$out/stderr_$c.log"
  python3 profiles/summarize_trace.py "$(find /tmp/pp -name '*kernel_trace.csv' | head -1)" 10 > "$out/per_step_${c}_${mode}_$sfx.txt"
  unset GV_BWD_SIDE GV_RGCN_BWD_SIDE GV_MADE_PREPARE
done
