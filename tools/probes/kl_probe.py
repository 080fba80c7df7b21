#!/usr/bin/env python3
"""gv_reparam_kl_fwd / _bwd (K3 + K6 fused: reparameterisation + KL to the mixture prior) at the bench sizes: us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib
from gcn_vae_amd.lib import ptr
from tools.microbench import timeit
for n in (14541, 40943):
    h, k = 200, 10
    dev = 'cuda'
    h2, eps = torch.randn(n, 2 * h, device=dev), torch.randn(n, h, device=dev)
    z_pre = torch.randn(2 * k, h, device=dev)
    z, v, m = (torch.empty(n, h, device=dev) for _ in range(3))
    resp = torch.empty(n, k, device=dev)
    ws = torch.empty(int(lib.load().gv_kl_workspace_bytes(n, h, k)) // 4, device=dev)
    gkl = torch.ones((), device=dev); gz_up = torch.randn(n, h, device=dev); gh2 = torch.empty(n, 2 * h, device=dev); gzp = torch.empty_like(z_pre)
    f = lambda: lib.call('gv_reparam_kl_fwd', ptr(h2), ptr(eps), ptr(z_pre), ptr(z), ptr(v), ptr(m), ptr(resp), ptr(ws), n, h, k, lib.stream())
    b = lambda: lib.call('gv_reparam_kl_bwd', ptr(z), ptr(h2), ptr(v), ptr(eps), ptr(z_pre), ptr(resp), ptr(gkl), 1.0, 0.01, ptr(gz_up), ptr(gh2),
                         ptr(gzp), 0, ptr(ws), n, h, k, lib.stream())
    f(); b()
    mb_f = n * h * 4 * (3 + 3) / 1e6 + n * k * 4 / 1e6
    mb_b = n * h * 4 * (1 + 2 + 1 + 1 + 1 + 2) / 1e6
    tf, tb = timeit(f), timeit(b)
    print(f'n={n}: reparam+KL fwd {tf:6.1f} us ({mb_f:.0f} MB, {mb_f / tf * 1e-6 * 1e6 / 1e3:.2f} TB/s)   bwd {tb:6.1f} us ({mb_b:.0f} MB, {mb_b / tb / 1e3:.2f} TB/s)')
    print('   checksum', float(resp.sum()), float(gh2.abs().sum()), float(gzp.abs().sum()))
