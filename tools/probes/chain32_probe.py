#!/usr/bin/env python3
"""gv_made_chain_f32 (one fp32 MADE pass, d = 200, 5 layers) by number of rows, with the masks' plan and with the dense plan, forward
and backward-x chain; us per launch (hipGraph replay) and the fraction of the fp32 MFMA peak the DENSE flop count is delivered at."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import made, ops
from gcn_vae_amd.flows import MADE
from tools.microbench import timeit

d = 200
torch.manual_seed(0)
mod = MADE(d, d, 3).cuda()
lin = mod._linears()
L = len(lin)
ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
bs = [l.bias.detach() for l in lin]
masks = [l.mask for l in lin]
widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
packed = made.made_pack_weights_f32(ws)
for m in [int(a) for a in sys.argv[1:]] or [64, 4096, 10240, 14741, 16384, 40943]:
    x = torch.randn(m, d, device='cuda')
    acts = [torch.empty(m, widths[l], device='cuda') for l in range(L)]
    grads = [torch.empty(m, widths[l], device='cuda') for l in range(L - 1)]
    g_top, gx = torch.randn(m, widths[-1], device='cuda'), torch.zeros(m, d, device='cuda')
    flops = 2.0 * m * sum(n * k for n, k in zip(widths, kin))
    for name, use in (('masks', True), ('dense', False)):
        pf = made.made_chain_f32_plan(widths, kin, masks if use else None)
        pb = made.made_chain_f32_plan(list(reversed(kin)), list(reversed(widths)), list(reversed(masks)) if use else None, transposed=True)
        fwd = lambda: made.made_chain_f32(x, m, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=acts[l])
                                                 for l in range(L)], pf)
        bwd = lambda: made.made_chain_f32(g_top, m, [dict(w_packed=packed[l][1], n=kin[l], k=widths[l], mask=acts[l - 1], out_f32=grads[l - 1])
                                                     for l in reversed(range(1, L))] +
                                          [dict(w_packed=packed[0][1], n=d, k=widths[0], out_f32=gx, accumulate=True)], pb)
        nost = lambda: made.made_chain_f32(x, m, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1,
                                                       out_f32=acts[l] if l == L - 1 else None) for l in range(L)], pf)
        tf, tb, tn = timeit(fwd), timeit(bwd), timeit(nost)
        print(f'm={m:6d} ({(m + 63) // 64:4d} tiles) {name}: forward {tf:7.1f} us ({flops / tf / 1e6 / 157:.2f} of 157 TF), '
              f'forward without hidden stores {tn:7.1f} us, backward-x {tb:7.1f} us ({flops / tb / 1e6 / 157:.2f})', flush=True)
