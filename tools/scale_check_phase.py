#!/usr/bin/env python3
"""BASELINE configs[4] scale on ONE GPU (1 M entities, 50 M directed edges, 2000 relation types, h = 200): K1 layer-1
forward / backward-x with the per-row kernels and with the relation-phase kernel (csrc/k_phase.hip)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops

n, e, r, nb, si, so = 1_000_000, 50_000_000, 2000, 100, 2, 2
gen = torch.Generator(device='cuda').manual_seed(0)
src = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
dst = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
et = torch.randint(0, r, (e,), device='cuda', generator=gen)
gidx = ops.GraphIndex(src, dst, n)
ridx = gidx.relation_index(et, r)
deg = torch.bincount(dst, minlength=n).float()
norm = (1.0 / deg.clamp(min=1))[dst]
w = torch.randn(r, nb * si * so, device='cuda', generator=gen)
x1 = torch.randn(n, nb * si, device='cuda', generator=gen)
by = e * (200 * 4 + 12) + n * (200 * 4 + 4) + r * 400 * 4

def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = fn()
    t.record(); torch.cuda.synchronize()
    return out, s.elapsed_time(t) / iters

for side, order, nbr, ety, tr in (('dst', gidx.by_dst, gidx.nbr_by_dst, ridx.et_by_dst, False),
                                  ('src', gidx.by_src, gidx.nbr_by_src, ridx.et_by_src, True)):
    ref, ms = timed(lambda: ops.bdd_aggregate(order.seg, nbr, ety, norm, order.perm, x1, w, nb, si, so, tr))
    print(f'{side} per-row : {ms:8.2f} ms  {by / ms / 1e6:7.1f} GB/s algorithmic ({by / ms / 1e6 / 8000:.2f} of 8 TB/s)', flush=True)
    t0 = time.time()
    ph = ridx.phase_order(gidx, side, nb, si, so)
    torch.cuda.synchronize()
    print(f'    phase index: {time.time() - t0:.1f} s, tiles {ph.n_tiles}, phases {ph.n_phases} x {ph.rels_per_phase}, rows/wave {ph.rows_per_wave}', flush=True)
    cp = ph.coef(norm)
    wp = ops.pack_weight_phase(ph, w, nb, si, so)
    got, ms = timed(lambda: ops.bdd_aggregate_phases(ph, cp, x1, wp, r, nb, si, so))
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f'{side} phases  : {ms:8.2f} ms  {by / ms / 1e6:7.1f} GB/s algorithmic ({by / ms / 1e6 / 8000:.2f} of 8 TB/s)  rel.diff {err:.1e}', flush=True)
    del ph, cp, got, ref
print('peak mem %.1f GiB' % (torch.cuda.max_memory_allocated() / 2**30))
