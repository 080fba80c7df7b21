# streamed phase kernel: column part by XCD (GV_PHASE_XCD) and two LDS weight buffers (GV_PHASE_BUFFERS=2), h = 500
for cfg in "GV_PHASE_XCD=1" "GV_PHASE_XCD=0" "GV_PHASE_XCD=1 GV_PHASE_BUFFERS=2" "GV_PHASE_XCD=1 GV_PHASE_LDS=122880"; do
  echo "== $cfg"
  env $cfg PHASE_BENCH_DEPTHS=0 PHASE_BENCH_STREAM_ONLY=1 timeout -k 10 300 python tools/phase_bench.py 500 2>&1 | grep "streamed" | cut -c1-130
done
