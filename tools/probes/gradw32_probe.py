#!/usr/bin/env python3
"""gv_made_gradw_f32 (200 x 200 weight gradient over K stacked rows, fp32) standalone: us per call (product + split sum), masked and
dense, beside the generic split-K GEMM it replaces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import made, ops
from gcn_vae_amd.flows import MADE
from tools.microbench import timeit

d = 200
mod = MADE(d, d, 3).cuda()
masks = [l.mask for l in mod._linears()]
for k in [int(a) for a in sys.argv[1:]] or [73705, 204715]:
    g, a = torch.randn(k, d, device='cuda'), torch.randn(k, d, device='cuda')
    g2 = torch.randn(k, 2 * d, device='cuda')
    out, db = torch.empty(d, d, device='cuda'), torch.empty(d, device='cuda')
    out2, db2 = torch.empty(2 * d, d, device='cuda'), torch.empty(2 * d, device='cuda')
    flops = 2.0 * k * d * d
    for name, mk in (('masked', masks[1]), ('dense', None)):
        t = timeit(lambda: made.made_gradw_f32(g, a, wmask=mk, out=out, db=db))
        print(f'k={k:7d} 200x200 {name}: {t:7.1f} us ({flops / t / 1e6:.1f} TF dense-equivalent, operands {2 * k * d * 4 / t / 1e6:.2f} TB/s)', flush=True)
    t = timeit(lambda: made.made_gradw_f32(g2, a, wmask=masks[-1], out=out2, db=db2))
    print(f'k={k:7d} 400x200 masked: {t:7.1f} us ({2 * flops / t / 1e6:.1f} TF dense-equivalent)', flush=True)
    t = timeit(lambda: ops.gemm(g, a, trans_a=True, split_k=ops.pick_split_k(d, d, k)))
    print(f'k={k:7d} 200x200 generic split-K GEMM: {t:7.1f} us ({flops / t / 1e6:.1f} TF)', flush=True)
