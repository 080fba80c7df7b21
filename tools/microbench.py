#!/usr/bin/env python3
"""Kernel micro-benchmarks on the C2 shapes (run on the GPU box):  python tools/microbench.py [gemm|k1|all]
Times each op with HIP events over many launches on torch's current stream; prints us per launch.
torch.mm (rocBLAS/hipBLASLt) is timed beside the hand-written f32-MFMA GEMM as a yardstick only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops  # noqa: E402


def timeit(fn, iters=50, warm=5):
    """GPU time per launch: the launches are captured into one hipGraph and replayed, so host-side launch
    cost (python + ctypes, ~10-15 us per call) cannot floor the figure."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if os.environ.get('MB_GRAPH', '1') == '1':
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / (3 * iters) * 1e3
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def bench_gemm():
    N = 14541
    shapes = [('fwd L1  x@Wl      NN', (N, 200, 200, False, False)), ('fwd L2  x@Wl      NN', (N, 400, 200, False, False)),
              ('bwd L1  g@Wl^T    NT', (N, 200, 200, False, True)), ('bwd L2  g@Wl^T    NT', (N, 200, 400, False, True)),
              ('bwd L1  x^T@g     TN', (200, 200, N, True, False)), ('bwd L2  x^T@g     TN', (200, 400, N, True, False))]
    for name, (m, n, k, ta, tb) in shapes:
        a = torch.randn((k, m) if ta else (m, k), device='cuda')
        b = torch.randn((n, k) if tb else (k, n), device='cuda')
        sk = ops.pick_split_k(m, n, k) if ta else 1
        t_mine = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=sk))
        t_bf16 = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, split_k=sk, precision='bf16'))
        aa, bb = (a.t() if ta else a), (b.t() if tb else b)
        t_torch = timeit(lambda: torch.mm(aa, bb))
        fl = 2.0 * m * n * k
        print(f'{name}  m={m:6d} n={n:4d} k={k:6d} split_k={sk:2d}: mine {t_mine:7.1f} us ({fl / t_mine / 1e6:6.1f} TF)   '
              f'bf16 operands {t_bf16:7.1f} us   torch.mm {t_torch:7.1f} us')


def bench_k1(chunk=256, chunk_rel=128):
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import FB15K237, synthetic_kg
    cfg = FB15K237
    data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
    g, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = g.edges()
    N, E, R = data.num_nodes, src.numel(), 2 * data.num_rels
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), N, chunk=chunk)
    ridx = ops.RelationIndex(gidx, torch.from_numpy(rel).cuda(), R, chunk=chunk_rel)
    norm = torch.from_numpy(node_norm).cuda()[dst.cuda()].contiguous()
    print(f'N={N} E={E} R={R} chunk={chunk}: items fwd {gidx.by_dst.seg.n_items} (split rows {gidx.by_dst.seg.n_fix}), '
          f'bwd {gidx.by_src.seg.n_items}, rel {ridx.by_rel.seg.n_items}')
    for (fin, fout) in ((200, 200), (200, 400)):
        nb = 100
        si, so = fin // nb, fout // nb
        x = torch.randn(N, fin, device='cuda')
        gg = torch.randn(N, fout, device='cuda')
        w = torch.randn(R, nb * si * so, device='cuda')
        pre = torch.randn(N, fout, device='cuda')
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x,
                                             w, nb, si, so, False, pre, 1))
        by = E * (fin * 4 + 12) + N * (fout * 4 + 4) + R * fin * fout // nb * 4
        print(f'agg fwd   {si}x{so}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic')
        wp = ops.pack_weight(w, nb, si, so)
        wpt = ops.pack_weight(w, nb, so, si, True)
        ref = ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x, w, nb, si, so, False, pre, 1)
        got = ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x, wp, nb, si, so, False, pre, 1, packed=True)
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, gidx.by_dst.perm, x,
                                             wp, nb, si, so, False, pre, 1, packed=True))
        print(f'agg fwd   {si}x{so}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic   [lane-packed W]  maxdiff {float((ref - got).abs().max()):.2e}')
        tp = timeit(lambda: ops.pack_weight(w, nb, si, so))
        print(f'pack W    {si}x{so}: {tp:7.1f} us')
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg,
                                             w, nb, so, si, True))
        by = E * (fout * 4 + 12) + N * (fin * 4 + 4) + R * fin * fout // nb * 4
        print(f'agg bwd-x {so}x{si}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic')
        norm_s = norm[gidx.by_src.perm.long()].contiguous()
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm_s, None, gg,
                                             w, nb, so, si, True))
        print(f'agg bwd-x {so}x{si}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic   [coefficients pre-permuted]')
        ref = ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg, w, nb, so, si, True)
        got = ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg, wpt, nb, so, si, True, packed=True)
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm, gidx.by_src.perm, gg,
                                             wpt, nb, so, si, True, packed=True))
        print(f'agg bwd-x {so}x{si}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic   [lane-packed W]  maxdiff {float((ref - got).abs().max()):.2e}')
        t = timeit(lambda: ops.bdd_grad_weight(ridx.by_rel.seg, ridx.src_by_rel, ridx.dst_by_rel, norm, ridx.by_rel.perm,
                                               x, gg, nb, si, so))
        by = E * (fin * 4 + fout * 4 + 12) + R * fin * fout // nb * 4
        print(f'grad-W    {si}x{so}: {t:7.1f} us  {by / t / 1e3:7.1f} GB/s algorithmic')


def probe_k1():
    """Where does K1's time go?  Re-times the forward/backward-x aggregations with (a) every edge on relation 0
    (weights then hit the per-CU cache), (b) every edge reading node 0 (features cached), (c) both."""
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import FB15K237, synthetic_kg
    cfg = FB15K237
    data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
    g, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = g.edges()
    N, R = data.num_nodes, 2 * data.num_rels
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), N)
    ridx = ops.RelationIndex(gidx, torch.from_numpy(rel).cuda(), R)
    norm = torch.from_numpy(node_norm).cuda()[dst.cuda()].contiguous()
    for (fin, fout) in ((200, 200), (200, 400)):
        nb = 100
        si, so = fin // nb, fout // nb
        pad = int(os.environ.get('MB_PAD', '0'))
        x = torch.randn(N, fin, device='cuda')
        gg = torch.randn(N, fout, device='cuda')
        if pad:                                  # rows start on `pad`-float boundaries (ld > row length)
            x = torch.randn(N, -(-fin // pad) * pad, device='cuda')[:, :fin]
            gg = torch.randn(N, -(-fout // pad) * pad, device='cuda')[:, :fout]
        w = torch.randn(R, nb * si * so, device='cuda')
        wp = ops.pack_weight(w, nb, si, so) if (ops.pack_supported(nb, si, so, False) and si * so >= 8) else None
        wpt = ops.pack_weight(w, nb, so, si, True) if (ops.pack_supported(nb, so, si, True) and si * so >= 8) else None
        for tag, et_d, nb_d, et_s, nb_s in (
                ('as is        ', ridx.et_by_dst, gidx.nbr_by_dst, ridx.et_by_src, gidx.nbr_by_src),
                ('one relation ', torch.zeros_like(ridx.et_by_dst), gidx.nbr_by_dst, torch.zeros_like(ridx.et_by_src), gidx.nbr_by_src),
                ('one neighbour', ridx.et_by_dst, torch.zeros_like(gidx.nbr_by_dst), ridx.et_by_src, torch.zeros_like(gidx.nbr_by_src)),
                ('both         ', torch.zeros_like(ridx.et_by_dst), torch.zeros_like(gidx.nbr_by_dst), torch.zeros_like(ridx.et_by_src),
                 torch.zeros_like(gidx.nbr_by_src))):
            tf = timeit(lambda: ops.bdd_aggregate(gidx.by_dst.seg, nb_d, et_d, norm, gidx.by_dst.perm, x,
                                                  wp if wp is not None else w, nb, si, so, False, None, 0, packed=wp is not None))
            tb = timeit(lambda: ops.bdd_aggregate(gidx.by_src.seg, nb_s, et_s, norm, gidx.by_src.perm, gg,
                                                  wpt if wpt is not None else w, nb, so, si, True, packed=wpt is not None))
            print(f'{si}x{so} {tag}: fwd {tf:6.1f} us   bwd-x {tb:6.1f} us')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'probe':
        probe_k1()
        sys.exit(0)
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('gemm', 'all'):
        bench_gemm()
    if what in ('k1', 'all'):
        bench_k1(*(int(a) for a in sys.argv[2:4]))
