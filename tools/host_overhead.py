#!/usr/bin/env python3
"""Host-side cost of one eagerly launched training step (python + ctypes + torch allocator + autograd engine), which bounds
the step rate whenever the step is not replayed as a hipGraph (multi-rank runs, mini-batches).  cProfile over N steps of
bench.py's workload with the GPU kept busy; prints the top functions by own time.
    python tools/host_overhead.py [steps] [--force-dist]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
steps = int(next((a for a in sys.argv[1:] if a.isdigit()), '200'))
import bench  # noqa: E402


def main():
    args = bench.parse.__wrapped__() if hasattr(bench.parse, '__wrapped__') else None
    from gcn_vae_amd.optim import FlatAdam
    ns = type('A', (), dict(gpus=1, steps=steps, warmup=5, hidden=200, n_bases=100, n_flows=0, positives=20000,
                            negative_sample=10, dropout=0.2, gemm_precision='f32'))()
    dev = torch.device('cuda', 0)
    w = bench.make_workload(0, 1, ns, dev)
    model = bench.build_model(w, ns, dev) if hasattr(bench, 'build_model') else None
    return w, model


if __name__ == '__main__':
    # bench.py keeps its setup inside main(); re-use it by running main() with --no-graph under the profiler
    sys.argv = ['bench.py', '--steps', str(steps), '--warmup', '10', '--no-graph', '--no-cpu-baseline', '--profile-steps', '0'] + \
        (['--force-dist'] if '--force-dist' in sys.argv else [])
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    bench.main()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)
    st.print_callers('_cuda_getDeviceCount')
    st.print_callers('is_available')
