"""f32 GEMM timing on given shapes:  python tools/gemm_shapes.py   (env GV_GEMM_RING / GV_GEMM_TILE select the kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microbench import timeit  # noqa: E402

from gcn_vae_amd import ops  # noqa: E402

N = 14541
for name, (m, n, k, ta, tb) in [('h500 fwd L1 NN', (N, 500, 500, 0, 0)), ('h500 fwd L2 NN', (N, 1000, 500, 0, 0)),
                                ('h500 bwd L1 NT', (N, 500, 500, 0, 1)), ('h500 bwd L2 NT', (N, 500, 1000, 0, 1)),
                                ('h500 bwd L1 TN', (500, 500, N, 1, 0)), ('h500 bwd L2 TN', (500, 1000, N, 1, 0))]:
    a = torch.randn((k, m) if ta else (m, k), device='cuda')
    b = torch.randn((n, k) if tb else (k, n), device='cuda')
    sk = ops.pick_split_k(m, n, k) if ta else 1
    t = timeit(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), split_k=sk))
    print(f'{name}: m={m} n={n} k={k} split_k={sk}: {t:7.1f} us  {2.0 * m * n * k / t / 1e6:6.1f} TF', flush=True)
