"""gv_gemm_f32_sparse against the dense product on the MADE(500, 500, 3) shapes: time per launch (hipGraph replay), blocks kept.
    python tools/probes/gemm_sparse.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from microbench import timeit  # noqa: E402

from gcn_vae_amd import ops  # noqa: E402
from gcn_vae_amd.flows import MADE  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 14541
m = MADE(500, 500, 3).cuda()
for li, lin in enumerate(m._linears()):
    if li not in (1, 4):
        continue
    mask = lin.mask
    w = (lin.weight * mask).detach().contiguous()
    o, i = w.shape
    x = torch.randn(rows, i, device='cuda')
    g = torch.randn(rows, o, device='cuda')
    fwd, bwd, tiles = (ops.block_words(mask, k) for k in ('fwd', 'bwd', 'tiles'))
    y = torch.empty(rows, o, device='cuda')
    gx = torch.empty(rows, i, device='cuda')
    t0 = timeit(lambda: ops.gemm(x, w, trans_b=True, out=y))
    t1 = timeit(lambda: ops.gemm(x, w, trans_b=True, out=y, b_k_chunks=fwd))
    print(f'layer {li} ({o} x {i}) forward  NT: dense {t0:7.1f} us, sparse {t1:7.1f} us', flush=True)
    t0 = timeit(lambda: ops.gemm(g, w, out=gx))
    t1 = timeit(lambda: ops.gemm(g, w, out=gx, b_k_chunks=bwd))
    print(f'layer {li} ({o} x {i}) backward NN: dense {t0:7.1f} us, sparse {t1:7.1f} us', flush=True)
    for sk in sorted({ops.pick_split_k(o, i, rows), 8, 16, 24, 32, 48, 64}):
        t0 = timeit(lambda: ops.gemm(g, x, trans_a=True, split_k=sk))
        t1 = timeit(lambda: ops.gemm(g, x, trans_a=True, split_k=sk, c_tiles=tiles))
        print(f'layer {li} ({o} x {i}) gradient TN (split_k {sk}{" = pick_split_k" if sk == ops.pick_split_k(o, i, rows) else ""}): dense {t0:7.1f} us, sparse {t1:7.1f} us', flush=True)

# floors: nothing wanted / one chunk per tile (what a launch costs apart from its blocks' loops)
li, lin = 1, m._linears()[1]
w = (lin.weight * lin.mask).detach().contiguous()
x = torch.randn(rows, 500, device='cuda')
g = torch.randn(rows, 500, device='cuda')
y = torch.empty(rows, 500, device='cuda')
none = torch.zeros(2, dtype=torch.int64, device='cuda')
one = torch.ones(8, dtype=torch.int64, device='cuda')
allk = torch.full((8,), -1, dtype=torch.int64, device='cuda')
half = torch.full((8,), 0xFFFF, dtype=torch.int64, device='cuda')
print('forward NT, all chunks via words: %.1f us; 16 of 32 chunks: %.1f us; one chunk: %.1f us' % (
    timeit(lambda: ops.gemm(x, w, trans_b=True, out=y, b_k_chunks=allk)), timeit(lambda: ops.gemm(x, w, trans_b=True, out=y, b_k_chunks=half)),
    timeit(lambda: ops.gemm(x, w, trans_b=True, out=y, b_k_chunks=one))), flush=True)
for sk in (1, 16):
    print('gradient TN split_k %d, no tile wanted: %.1f us' % (sk, timeit(lambda: ops.gemm(g, x, trans_a=True, split_k=sk, c_tiles=none))), flush=True)
