# configs[4] on one GPU: the K1 launches of the bench step with the streamed / batch-per-list phase kernels and ring depths
for cfg in "GV_PHASE_STREAM=0" "GV_PHASE_STREAM=1" "GV_PHASE_STREAM=1 GV_PHASE_STREAM_D=4" "GV_PHASE_STREAM=1 GV_PHASE_STREAM_D=8"; do
  echo "== $cfg"
  env $cfg python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-check --repeats 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step ms', round(d['ms_per_step'],2), {k:(round(v['avg_us']),v.get('frac_of_hbm_peak')) for k,v in d['roofline_detail'].items() if k.startswith('agg_') and '1x1' not in k})"
done
