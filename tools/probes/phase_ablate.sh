# streamed phase kernel with parts switched off (GV_PHASE_DEBUG: 1 no barriers, 2 no staging; results are wrong in those modes)
for dbg in 0 1 2 3; do
  echo "== GV_PHASE_DEBUG=$dbg"
  GV_PHASE_DEBUG=$dbg PHASE_BENCH_DEPTHS=0 PHASE_BENCH_STREAM_ONLY=1 timeout -k 10 300 python tools/phase_bench.py 500 2>&1 | grep -v amdgpu.ids
done
