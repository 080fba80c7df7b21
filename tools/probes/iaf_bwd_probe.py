#!/usr/bin/env python3
"""gv_iaf_update_bwd_bf16_ex at WN18RR size (the middle passes' form: g_z accumulated, g_mu half, tiled transposed copy), whole
rows and one row block: us per launch over operands that are not cache-resident (four operand sets in turn)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib
from gcn_vae_amd.lib import ptr
from tools.microbench import timeit

d, SETS = 200, 4
dev = torch.device('cuda:0')
bf = dict(dtype=torch.bfloat16, device=dev)
cc = torch.ones(d, dtype=torch.int32, device=dev); cc[-1] = 0
for n in (40943, 20480):
    T = (n + 63) // 64
    sets = [dict(z=torch.randn(n, d, device=dev), ex=torch.rand(n, d, device=dev) + 0.5, gx=torch.randn(n, d, device=dev),
                 gz=torch.zeros(n, d, device=dev), gb=torch.empty(n, d, **bf), gt=torch.empty(T, 2 * d, 64, **bf)) for _ in range(SETS)]
    def run():
        for s in sets:
            lib.call('gv_iaf_update_bwd_bf16_ex', ptr(s['z']), ptr(s['ex']), d, ptr(cc), ptr(s['gx']), None, ptr(s['gz']), ptr(s['gb']), d,
                     ptr(s['gt']), 2 * d * 64, None, 2 | 4, n, d, lib.stream())
    us = timeit(run, iters=10) / SETS
    mb = n * d * (4 * 4 + 4 + 2 + 4) / 1e6          # reads z, ex, gx, g_z; writes g_z, the g_mu half, both transposed halves
    print(f'n={n}: {us:6.1f} us, {mb:.0f} MB -> {mb / us:.2f} TB/s', flush=True)
