#!/usr/bin/env python3
"""K1 by relation phases (csrc/k_phase.hip) beside the per-row kernels on the FB15k-237-shaped graph:
    python tools/phase_bench.py [hidden ...]          (default 200 500)
prints us per launch, algorithmic GB/s (SURVEY 8(d) bytes) and the largest difference between the two paths."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops, sampling  # noqa: E402
from gcn_vae_amd.data import FB15K237, synthetic_kg  # noqa: E402
from tools.microbench import timeit  # noqa: E402


def main():
    hiddens = [int(a) for a in sys.argv[1:]] or [200, 500]
    cfg = FB15K237
    data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
    g, rel, node_norm = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
    src, dst = g.edges()
    N, E, R = data.num_nodes, src.numel(), 2 * data.num_rels
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), N)
    ridx = ops.RelationIndex(gidx, torch.from_numpy(rel).cuda(), R)
    norm = torch.from_numpy(node_norm).cuda()[dst.cuda()].contiguous()
    nb = 100
    for h in hiddens:
        for fin, fout in ((h, h), (h, 2 * h)):
            si, so = fin // nb, fout // nb
            x = torch.randn(N, fin, device='cuda')
            gg = torch.randn(N, fout, device='cuda')
            w = torch.randn(R, nb * si * so, device='cuda') * 0.1
            pre = torch.randn(N, fout, device='cuda')
            pre_in = torch.randn(N, fin, device='cuda')
            for side, feat, p, q, tr, add in (('dst', x, si, so, False, pre), ('src', gg, so, si, True, pre_in)):
                order = gidx.by_dst if side == 'dst' else gidx.by_src
                nbr = gidx.nbr_by_dst if side == 'dst' else gidx.nbr_by_src
                ety = ridx.et_by_dst if side == 'dst' else ridx.et_by_src
                by = E * (nb * p * 4 + 12) + N * (nb * q * 4 + 4) + R * nb * p * q * 4
                pk = p * q >= 8 and ops.pack_supported(nb, p, q, tr)
                wk = ops.pack_weight(w, nb, p, q, tr) if pk else w
                coef_o = norm if order.perm is None else norm[order.perm.long()].contiguous()
                ref = ops.bdd_aggregate(order.seg, nbr, ety, coef_o, None, feat, wk, nb, p, q, tr, add, 1 if add is not None else 0, packed=pk)
                t_row = timeit(lambda: ops.bdd_aggregate(order.seg, nbr, ety, coef_o, None, feat, wk, nb, p, q, tr, add,
                                                         1 if add is not None else 0, packed=pk))
                stream_only = os.environ.get('PHASE_BENCH_STREAM_ONLY') == '1'      # geometries only the streamed kernel has
                os.environ['GV_PHASE_STREAM'] = '1' if stream_only else '0'
                tl = ridx.phase_order(gidx, side, nb, p, q)                        # the batch-per-list kernel's geometry
                if tl is None:
                    print(f'h={h} {side} {p}x{q}: per-row {t_row:7.1f} us; no phase kernel')
                    continue
                ct = tl.coef(norm)
                wp = ops.pack_weight_phase(tl, w, nb, p, q)
                got = ops.bdd_aggregate_phases(tl, ct, feat, wp, R, nb, p, q, add, 1 if add is not None else 0)
                err = float((got - ref).abs().max() / ref.abs().max())
                t_tile = timeit(lambda: ops.bdd_aggregate_phases(tl, ct, feat, wp, R, nb, p, q, add, 1 if add is not None else 0))
                t_pack = timeit(lambda: ops.pack_weight_phase(tl, w, nb, p, q))
                print(f'h={h} {side} {p}x{q}: per-row {t_row:7.1f} us ({by / t_row / 1e3:6.0f} GB/s)   phases {t_tile:7.1f} us '
                      f'({by / t_tile / 1e3:6.0f} GB/s = {by / t_tile / 1e3 / 8000:.2f} of 8 TB/s) + pack {t_pack:5.1f} us   '
                      f'rel.diff {err:.1e}   [tiles {tl.n_tiles}, phases {tl.n_phases} x {tl.rels_per_phase} relations, '
                      f'{tl.rows_per_wave} rows/wave, threads {tl.threads}]', flush=True)
                # the streamed form (csrc/k_stream.hip) in ITS geometry: the same per-row order, so bit-identical sums; ring depths GV_PHASE_STREAM_D
                os.environ['GV_PHASE_STREAM'] = '1'
                ts = ridx.phase_order(gidx, side, nb, p, q)
                cs = ts.coef(norm)
                for d in DEPTHS:
                    os.environ['GV_PHASE_STREAM_D'] = str(d)
                    got_s = ops.bdd_aggregate_phases(ts, cs, feat, wp, R, nb, p, q, add, 1 if add is not None else 0)
                    same = bool(torch.equal(got_s, got))
                    t_s = timeit(lambda: ops.bdd_aggregate_phases(ts, cs, feat, wp, R, nb, p, q, add, 1 if add is not None else 0))
                    print(f'    streamed, depth {d or "default"}: {t_s:7.1f} us ({by / t_s / 1e3:6.0f} GB/s = {by / t_s / 1e3 / 8000:.2f} '
                          f'of 8 TB/s)   bit-identical: {same}   [tiles {ts.n_tiles}, {ts.rows_per_wave} rows/wave, threads {ts.threads}]',
                          flush=True)
                os.environ.pop('GV_PHASE_STREAM_D', None)


DEPTHS = [int(d) for d in os.environ.get('PHASE_BENCH_DEPTHS', '0,4,6,8').split(',')]

if __name__ == '__main__':
    main()
