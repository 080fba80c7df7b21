#!/usr/bin/env python3
"""Where the fixed cost of gv_made_chain_f32 sits: chains of 1..5 layers (200 wide) on ONE workgroup (m = 64) and on a full chip (m = 16384)."""
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
ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
bs = [l.bias.detach() for l in lin]
masks = [l.mask for l in lin]
packed = made.made_pack_weights_f32(ws)
for m in (64, 16384):
    x = torch.randn(m, d, device='cuda')
    for L in (1, 2, 3, 4):
        widths, kin = [d] * L, [d] * L
        out = torch.empty(m, d, device='cuda')
        for name, use in (('masks', True), ('dense', False)):
            pf = made.made_chain_f32_plan(widths, kin, masks[:L] if use else None)
            fn = lambda: made.made_chain_f32(x, m, [dict(w_packed=packed[l][0], n=d, k=d, bias=bs[l], relu=True,
                                                         out_f32=out if l == L - 1 else None) for l in range(L)], pf)
            print(f'm={m:6d} layers={L} {name}: {timeit(fn):7.1f} us', flush=True)
