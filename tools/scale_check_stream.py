#!/usr/bin/env python3
"""BASELINE configs[4] scale on ONE GPU (1 M entities, 50 M directed edges, 2000 relation types, h = 200): K1 forward / backward-x of
both layers (2x2 and 2x4 / 4x2 blocks) on the two relation-phase kernels -- csrc/k_phase.hip (GV_PHASE_STREAM=0) and the streamed
csrc/k_stream.hip -- over the same lists.   python tools/scale_check_stream.py [2x2] [2x4]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops

n, e, r, nb = 1_000_000, 50_000_000, 2000, 100
SHAPES = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(2, 2), (2, 4)]
gen = torch.Generator(device='cuda').manual_seed(0)
src = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
dst = (torch.rand(e, device='cuda', generator=gen) ** 2 * n).long().clamp_(max=n - 1)
et = torch.randint(0, r, (e,), device='cuda', generator=gen)
gidx = ops.GraphIndex(src, dst, n)
ridx = gidx.relation_index(et, r)
deg = torch.bincount(dst, minlength=n).float()
norm = (1.0 / deg.clamp(min=1))[dst]

def timed(fn, iters=3):
    fn(); torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): out = fn()
    t.record(); torch.cuda.synchronize()
    return out, s.elapsed_time(t) / iters

def run_shape(si, so):
    w = torch.randn(r, nb * si * so, device='cuda', generator=gen)
    x_in = torch.randn(n, nb * si, device='cuda', generator=gen)
    g_out = torch.randn(n, nb * so, device='cuda', generator=gen)
    for side, feat, p, q, tr in (('dst', x_in, si, so, False), ('src', g_out, so, si, True)):
        by = e * (nb * p * 4 + 12) + n * (nb * q * 4 + 4) + r * nb * p * q * 4
        res = {}
        for stream in ('0', '1'):
            os.environ['GV_PHASE_STREAM'] = stream
            ph = ridx.phase_order(gidx, side, nb, p, q)
            cp = ph.coef(norm)
            wp = ops.pack_weight_phase(ph, w, nb, p, q)
            got, ms = timed(lambda: ops.bdd_aggregate_phases(ph, cp, feat, wp, r, nb, p, q))
            res[stream] = got
            print(f'{side} {p}x{q} {"streamed" if stream == "1" else "phases  "}: {ms:8.2f} ms  {by / ms / 1e6:7.1f} GB/s algorithmic '
                  f'({by / ms / 1e6 / 8000:.2f} of 8 TB/s)   [tiles {ph.n_tiles}, phases {ph.n_phases} x {ph.rels_per_phase}, '
                  f'{ph.rows_per_wave} rows/wave]', flush=True)
        print(f'    bit-identical: {bool(torch.equal(res["0"], res["1"]))}', flush=True)
        del res, got


for si, so in SHAPES:
    run_shape(si, so)
print('peak mem %.1f GiB' % (torch.cuda.max_memory_allocated() / 2**30))
