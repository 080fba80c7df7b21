#!/usr/bin/env python3
"""K1 at BASELINE configs[2]'s shape (WN18RR-sized synthetic: 40 943 entities, 22 directed relation types, 173 670 directed
edges, num_bases = 20 -> 10x10 / 10x20 / 20x10 blocks): the LDS-resident kernel (csrc/k_lds.hip) beside the per-row kernels,
with probes that say where its time goes.   python tools/k1_lds_bench.py [probe]   (run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gcn_vae_amd import ops, sampling  # noqa: E402
from gcn_vae_amd.data import synthetic_kg  # noqa: E402
from microbench import timeit  # noqa: E402


def main(probe):
    N, NR, T, nb = 40943, 11, 86835, 20
    data = synthetic_kg(N, NR, T, seed=0)
    g, rel, node_norm = sampling.build_test_graph(N, NR, data.train)
    src, dst = g.edges()
    R, E = 2 * NR, src.numel()
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), N)
    ridx = ops.RelationIndex(gidx, torch.from_numpy(rel).cuda(), R)
    norm = torch.from_numpy(node_norm).cuda()[dst.cuda()].contiguous()
    norm_s = norm[gidx.by_src.perm.long()].contiguous()
    chunks = [int(c) for c in os.environ.get('LB_CHUNKS', '64').split(',')]      # most edges of a super-item
    for fin, fout in ((200, 200), (200, 400)):
        si, so = fin // nb, fout // nb
        x = torch.randn(N, fin, device='cuda')
        gg = torch.randn(N, fout, device='cuda')
        w = torch.randn(R, nb * si * so, device='cuda') * 0.1
        pre = torch.randn(N, fout, device='cuda')
        keep = (torch.rand(N, fout, device='cuda') > 0.2).to(torch.uint8)
        bf = os.environ.get('LB_BF16', '0') == '1'
        pf, pb = ops.lds_plan(R, nb, si, so, bf=bf), ops.lds_plan(R, nb, so, si, bf=bf)
        wf, wb = ops.pack_weight_lds(w, nb, si, so, False, pf), ops.pack_weight_lds(w, nb, so, si, True, pb)
        by_f = E * (fin * 4 + 12) + N * (fout * 4 + 4) + R * fin * fout // nb * 4
        by_b = E * (fout * 4 + 12) + N * (fin * 4 + 4) + R * fin * fout // nb * 4
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, norm, None, x, w, nb, si, so, False, pre, 1, keep, 1.25))
        print(f'{si}x{so} fwd   per-row kernels (chunk 256): {t:7.1f} us  {by_f / t / 1e3:7.1f} GB/s')
        t = timeit(lambda: ops.bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, norm_s, None, gg, w, nb, so, si, True))
        print(f'{so}x{si} bwd-x per-row kernels (chunk 256): {t:7.1f} us  {by_b / t / 1e3:7.1f} GB/s')
        for chunk in chunks:
            sd, ss = gidx.lds_order('dst', chunk), gidx.lds_order('src', chunk)
            variants = [('as is', gidx.nbr_by_dst, ridx.et_by_dst, gidx.nbr_by_src, ridx.et_by_src)]
            if probe:
                z = torch.zeros_like
                variants += [('one neighbour', z(gidx.nbr_by_dst), ridx.et_by_dst, z(gidx.nbr_by_src), ridx.et_by_src),
                             ('one relation', gidx.nbr_by_dst, z(ridx.et_by_dst), gidx.nbr_by_src, z(ridx.et_by_src))]
            for tag, nd, ed, ns, es in variants:
                tf = timeit(lambda: ops.bdd_aggregate_lds(sd, nd, ed, norm, None, x, wf, R, nb, si, so, False, pre, 1, keep, 1.25, plan=pf))
                tb = timeit(lambda: ops.bdd_aggregate_lds(ss, ns, es, norm_s, None, gg, wb, R, nb, so, si, True, plan=pb))
                print(f'{si}x{so} LDS-resident{" bf16" if bf else ""} ({pf[0]}/{pb[0]} parts) <= {chunk:3d} edges [{tag:13s}] super-items {sd.n_sitems}/{ss.n_sitems} split rows {sd.n_fix}/{ss.n_fix} empty {sd.n_empty}/{ss.n_empty}: '
                      f'fwd {tf:6.1f} us {by_f / tf / 1e3:7.1f} GB/s   bwd-x {tb:6.1f} us {by_b / tb / 1e3:7.1f} GB/s')
        tp = timeit(lambda: ops.pack_weight_lds(w, nb, si, so, False, pf))
        print(f'pack W {si}x{so}: {tp:5.1f} us')


if __name__ == '__main__':
    main(len(sys.argv) > 1 and sys.argv[1] == 'probe')
