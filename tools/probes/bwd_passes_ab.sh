cd "$(dirname "$0")/../.."
one() { tag=$1; shift; timeout -k 10 200 python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', 'ms', round(d['ms_per_step'],4), flush=True)"; }
for i in 1 2; do
for pl in 1 2 3; do GV_MADE_CHAIN_PASSES=$pl one "c3 bwd_passes=$pl" --config c3; done
GV_MADE_ROW_BLOCKS=3 one "c3 blocks=3" --config c3
done
for pl in 1 2 3; do GV_MADE_CHAIN_PASSES=$pl one "c2f3 bwd_passes=$pl" --n-flows 3 --gemm-precision bf16; done
