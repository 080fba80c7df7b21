set -x
PHASE_BENCH_DEPTHS=0 timeout -k 10 300 python tools/phase_bench.py 500 > gpurun_out/r5_pb_base.log 2>&1
GV_PHASE_ROWS=16 GV_PHASE_THREADS=512 PHASE_BENCH_DEPTHS=4,6,8 PHASE_BENCH_STREAM_ONLY=1 timeout -k 10 300 python tools/phase_bench.py 500 > gpurun_out/r5_pb_k16.log 2>&1
grep -v "^+" gpurun_out/r5_pb_k16.log | tail -30
