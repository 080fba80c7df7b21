cd "$(dirname "$0")/../.."
one() { tag=$1; v=$2; shift 2; GV_GRADW_SPLIT_MAX_SIDE=$v timeout -k 10 200 python bench.py "$@" --steps 40 --warmup 10 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag split', $v, 'ms', round(d['ms_per_step'],4), flush=True)"; }
for i in 1 2 3; do for v in 96 128; do one c3 $v --config c3; done; done
for i in 1 2; do for v in 96 128; do one c2f3bf16 $v --n-flows 3 --gemm-precision bf16; done; done
