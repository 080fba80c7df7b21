#!/usr/bin/env python3
"""In-kernel timeline of the bf16 gv_made_chain (gv_made_chain_debug_stamps): s_memtime stamps of workgroup 0's eight waves for the
last forward chain (with the fused IAF update) and the last backward chain of one bf16 MADE node at WN18RR size.
Per unit: wait = fragment fence, mma, epi = fragment issue + epilogue; 'gap' = layer barrier + store wave."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib, ops
from gcn_vae_amd.flows import MADE
from gcn_vae_amd.lib import ptr

m, d = int(sys.argv[1]) if len(sys.argv) > 1 else 40943, 200
torch.manual_seed(0)
mod = MADE(d, d, 3).cuda()
z = torch.randn(m, d, device='cuda', requires_grad=True)
buf = torch.zeros(8 * 64, dtype=torch.int32, device='cuda')


def show(tag):
    torch.cuda.synchronize()
    ts = buf.cpu().numpy().astype('int64').reshape(8, 64) & 0xffffffff
    t0 = ts[:, 0].min()
    print(f'--- {tag} (cycles from the first wave\'s start; 7 MMA waves + store wave 7)')
    for w in range(8):
        row = ts[w]
        n = int((row != 0).sum())
        rel = [(int(v) - int(t0)) & 0xffffffff for v in row[:n]]
        units = []
        for b in range(2, n - 1, 4):
            if b + 3 < n:
                units.append(f'[enter {rel[b]} wait {rel[b + 1] - rel[b]} mma {rel[b + 2] - rel[b + 1]} epi {rel[b + 3] - rel[b + 2]} end {rel[b + 3]}]')
        print(f'wave {w}: start {rel[0]} loop {rel[1] if n > 1 else -1} ' + ' '.join(units) + f' exit {rel[n - 1]}')
    buf.zero_()


with ops.gemm_precision('bf16'):
    for _ in range(2):
        x, ld = mod(z)
        (x.sum() + ld.sum()).backward()
    torch.cuda.synchronize()
    lib.call('gv_made_chain_debug_stamps', ptr(buf))
    x, ld = mod(z)
    show('last forward chain (pass 5, IAF update fused)')
    (x.sum() + ld.sum()).backward()
    show('last backward chain (pass 1)')
    lib.call('gv_made_chain_debug_stamps', None)
