#!/usr/bin/env python3
"""BASELINE configs[4] scale on ONE GPU: 1 M entities, 50 M directed edges, 2000 relation types, h=200, B=100.
Builds the device index, runs K1 forward / backward-x / grad-W of layer 1, checks linearity and times them
(the feature table is 800 MB here: a true HBM test, unlike FB15k-237)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops

n, e, r, nb, si, so = 1_000_000, 50_000_000, 2000, 100, 2, 2
gen = torch.Generator(device='cuda').manual_seed(0)
src = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
dst = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
et = torch.randint(0, r, (e,), device='cuda', generator=gen)
torch.cuda.synchronize(); t0 = time.time()
gidx = ops.GraphIndex(src, dst, n)
ridx = gidx.relation_index(et, r)
torch.cuda.synchronize(); print(f'index build: {time.time() - t0:.2f} s; items fwd {gidx.by_dst.seg.n_items} (split rows {gidx.by_dst.seg.n_fix}), '
                                f'by-rel {ridx.by_rel.seg.n_items}; mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB')
deg = torch.bincount(dst, minlength=n).float()
norm = (1.0 / deg.clamp(min=1))[dst]
w = torch.randn(r, nb * si * so, device='cuda', generator=gen)
x1 = torch.randn(n, nb * si, device='cuda', generator=gen)
x2 = torch.randn(n, nb * si, device='cuda', generator=gen)

def timed(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = fn()
    t.record(); torch.cuda.synchronize()
    return out, s.elapsed_time(t) / iters

agg = lambda xx: ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, xx, w, nb, si, so)
a1, ms = timed(lambda: agg(x1))
by = e * (200 * 4 + 12) + n * (200 * 4 + 4) + r * 400 * 4
print(f'K1 forward : {ms:8.2f} ms  {by / ms / 1e6:7.1f} GB/s algorithmic ({by / 1e9:.1f} GB)')
a2 = agg(x2); a12 = agg(2.0 * x1 - 0.5 * x2)
err = float((a12 - (2.0 * a1 - 0.5 * a2)).abs().max()); scale = float(a1.abs().max())
print(f'linearity  : max |agg(2x1-.5x2) - (2agg(x1)-.5agg(x2))| = {err:.3e} (scale {scale:.2f})')
assert err < 1e-4 * max(1.0, scale)
g = torch.randn(n, nb * so, device='cuda', generator=gen)
_, ms = timed(lambda: ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, g, w, nb, so, si, True))
print(f'K1 backward-x: {ms:8.2f} ms  {by / ms / 1e6:7.1f} GB/s algorithmic')
gw, ms = timed(lambda: ops.bdd_grad_weight(ridx.by_rel.seg, ridx.src_by_rel, ridx.dst_by_rel, norm, ridx.by_rel.perm, x1, g, nb, si, so))
byw = e * (400 * 4 + 12) + r * 400 * 4
print(f'K1 grad-W  : {ms:8.2f} ms  {byw / ms / 1e6:7.1f} GB/s algorithmic')
# grad-W check on one relation against torch
rel = 7
sel = (et == rel).nonzero().view(-1)[:200000]
ref = torch.einsum('e,ebi,ebj->bij', norm[sel].double(), x1[src[sel]].view(-1, nb, si).double(), g[dst[sel]].view(-1, nb, so).double())
if sel.numel() == int((et == rel).sum()):
    assert float((gw[rel].view(nb, si, so).double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())
    print('grad-W relation check ok')
print('peak mem %.1f GiB' % (torch.cuda.max_memory_allocated() / 2**30))
