#!/usr/bin/env python3
"""Raw-MRR evaluation throughput: the whole FB15k-237 test split in both directions
(2 x 20 466 queries against 14 541 entities, h = 200) through the fused rank-count scorer (gv_rank_scores) and through the
materialised form (one GEMM per 100-query batch + torch sigmoid / gather / compare / sum -- what ranking.py did before).
    python tools/eval_bench.py [--cpu-rows 200]     # --cpu-rows: also time the reference's (h, Eb, V) formulation on the host"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ranking   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu-rows', type=int, default=0)
    args = ap.parse_args()
    gen = torch.Generator().manual_seed(0)
    v, h, n, n_rel = 14541, 200, 20466, 237
    emb = (torch.randn(v, h, generator=gen) * 0.3).cuda()
    w = torch.randn(n_rel, h, generator=gen).cuda()
    trip = torch.stack([torch.randint(0, v, (n,), generator=gen), torch.randint(0, n_rel, (n,), generator=gen),
                        torch.randint(0, v, (n,), generator=gen)], 1).cuda()

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t_f, mrr_f = timed(lambda: ranking.calc_mrr(emb, w, trip, hits=[1, 3, 10], eval_bz=100, verbose=False), 5)
    fused = ranking.perturb_and_get_rank
    ranking.perturb_and_get_rank = ranking.perturb_and_get_rank_unfused
    try:
        t_u, mrr_u = timed(lambda: ranking.calc_mrr(emb, w, trip, hits=[1, 3, 10], eval_bz=100, verbose=False), 2)
    finally:
        ranking.perturb_and_get_rank = fused
    flop = 2 * 2.0 * n * v * h * 2          # two directions, two passes (target probability, count)
    print(f'fused    : {t_f * 1e3:8.2f} ms per full evaluation ({2 * n} queries)  MRR {mrr_f:.6f}  '
          f'{flop / t_f / 1e12:.1f} TFLOP/s f32 MFMA (two passes)')
    print(f'unfused  : {t_u * 1e3:8.2f} ms (410 batches of 100: GEMM + sigmoid + gather + compare + sum)  MRR {mrr_u:.6f}')
    if args.cpu_rows:
        e, ww = emb.cpu(), w.cpu()
        s, r, o = (trip[:args.cpu_rows, i].cpu() for i in range(3))
        t0 = time.perf_counter()
        emb_ar = (e[s] * ww[r]).transpose(0, 1).unsqueeze(2)            # (h, Eb, 1)   kgvae/utils.py:195-203
        emb_c = e.transpose(0, 1).unsqueeze(1)                          # (h, 1, V)
        score = torch.sigmoid(torch.sum(torch.bmm(emb_ar, emb_c), dim=0))
        _, idx = torch.sort(score, dim=1, descending=True)
        ranks = torch.nonzero(idx == o.view(-1, 1))[:, 1]
        dt = time.perf_counter() - t0
        print(f'host, reference formulation: {dt * 1e3:.1f} ms for {args.cpu_rows} queries -> '
              f'{dt / args.cpu_rows * 2 * n:.1f} s per full evaluation ({torch.get_num_threads()} threads)')


if __name__ == '__main__':
    main()
