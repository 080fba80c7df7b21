# two independent 512-thread workgroups per CU (half the LDS each): one computes while the other waits at its phase barrier
for cfg in "GV_PHASE_THREADS=512 GV_PHASE_LDS=80000" "GV_PHASE_THREADS=512 GV_PHASE_LDS=81920 GV_PHASE_BUFFERS=1" "GV_PHASE_THREADS=256 GV_PHASE_LDS=40000"; do
  echo "== $cfg"
  env $cfg PHASE_BENCH_DEPTHS=0 PHASE_BENCH_STREAM_ONLY=1 timeout -k 10 300 python tools/phase_bench.py 500 2>&1 | grep "streamed" | cut -c1-160
done
