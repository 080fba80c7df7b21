#!/bin/bash
# the bench lines of the secondary regimes (mini-batch, flows, bf16): tools/collect_round5_lines.sh  -> gpurun_out/r5/bench_<tag>.json
out=$PWD/gpurun_out/r5; mkdir -p "$out"
run() { tag=$1; shift; echo "== $tag"; timeout -k 10 600 python bench.py "$@" > "$out/bench_$tag.json" 2> "$out/bench_$tag.err" || echo "bench $tag rc=$?"; }
run mb --config mb
run mb_flows3 --config mb --n-flows 3
run mb_flows3_bf16 --config mb --n-flows 3 --gemm-precision bf16
run c2_flows3_bf16 --config c2 --n-flows 3 --gemm-precision bf16
run mb_h500_flows3 --config mb --hidden 500 --n-flows 3 --steps 20
python tools/show_bench.py "$out"/bench_mb*.json "$out"/bench_c2_flows3_bf16.json 2>/dev/null | grep json
