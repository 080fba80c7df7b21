# A/B of the K slices of the side-stream weight-gradient products: bash tools/probes/gradw_slices_ab.sh
for v in 96 256 96 256; do GV_GRADW_SPLIT_MAX_SIDE=$v timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline --no-check > gpurun_out/ab_c3_s$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ab_c3_s$v.json').read().strip().splitlines()[-1]); print('c3 side gradw slices', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
for v in 96 256 96 256; do GV_GRADW_SPLIT_MAX_SIDE=$v timeout -k 10 300 python bench.py --n-flows 3 --gemm-precision bf16 --no-cpu-baseline --no-check > gpurun_out/ab_c2f_s$v.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ab_c2f_s$v.json').read().strip().splitlines()[-1]); print('c2f side gradw slices', $v, d['ms_per_step'], d.get('ms_per_step_median'))"; done
