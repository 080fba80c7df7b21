#!/bin/bash
# per-dispatch durations of the index-builder kernels inside the replayed mini-batch step (last step of the trace)
mkdir -p gpurun_out/r3mb && export TMPDIR=/tmp && rm -rf /tmp/pp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o t -- python3 bench.py --config mb --steps 10 --warmup 3 --no-cpu-baseline --no-check --profile-steps 0 > gpurun_out/r3mb/bench_prof.json 2> gpurun_out/r3mb/err.log || exit 1
f=$(find /tmp/pp -name "*kernel_trace.csv" | head -1)
python3 profiles/summarize_trace.py "$f" 10 > gpurun_out/r3mb/per_step_summary_mb.txt
head -1 $f > gpurun_out/r3mb/idx_rows.csv
grep "k_rs_\|k_items_blocks\|k_lower_bounds\|k_gather\|k_triplet" $f | tail -60 >> gpurun_out/r3mb/idx_rows.csv
head -1 gpurun_out/r3mb/per_step_summary_mb.txt
