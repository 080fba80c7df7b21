#!/usr/bin/env python3
"""Predicted multi-GPU step time of the two partition schemes, from byte counts and the xGMI figures of
SURVEY.md 8(e): 7 links x 153 GB/s per GPU, point to point.  No measurement is involved: this is the yardstick the first measured
SCALE_rNN.json is to be judged against.   python tools/scaling_model.py [--eff 0.8] [--alpha-us 12]"""
import argparse

CONFIGS = {
    # name: (N, h, arena MB besides the entity table, measured 1-GPU step ms, share of the step that is node-level (replicated under
    #        edge sharding: self-loop GEMMs, reparam, KL, MMD, Adam), E per rank)
    'c2 (FB15k-237, h=200)': dict(N=14541, h=200, other_mb=2.9, t1_ms=1.12, node_share=0.45),
    'c4 (FB15k-237, h=500)': dict(N=14541, h=500, other_mb=14.6, t1_ms=4.15, node_share=0.35),
}


def t_allreduce(S, p, bw, alpha, direct):
    """S bytes all-reduced over p ranks.  ring: 2(p-1)/p * S over ONE link; direct: reduce-scatter + all-gather, every rank
    exchanging S/p with each peer on its own link."""
    if p == 1:
        return 0.0
    if direct:
        return 2 * (S / p) / bw + 2 * alpha
    return 2 * (p - 1) / p * S / bw + 2 * (p - 1) * alpha


def t_gather(S, p, bw, alpha, direct):
    """all-gather (or reduce-scatter) of S bytes total, S/p per rank."""
    if p == 1:
        return 0.0
    if direct:
        return (S / p) / bw + alpha
    return (p - 1) / p * S / bw + (p - 1) * alpha


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--eff', type=float, default=0.8, help='achieved fraction of the 153 GB/s link peak')
    ap.add_argument('--alpha-us', type=float, default=12.0, help='per-hop / per-phase latency of a collective')
    a = ap.parse_args()
    bw, alpha = 153e9 * a.eff, a.alpha_us * 1e-6
    for name, c in CONFIGS.items():
        S1, S2 = c['N'] * c['h'] * 4, c['N'] * 2 * c['h'] * 4
        table, other = S1, c['other_mb'] * 1e6
        print(f'\n{name}: S1 = {S1 / 1e6:.1f} MB, S2 = {S2 / 1e6:.1f} MB, gradient arena = {(table + other) / 1e6:.1f} MB, '
              f'1-GPU step {c["t1_ms"]} ms (weak scaling: per-rank edges fixed)')
        print(f'{"ranks":>5} {"scheme":>22} {"collective ms":>14} {"hidden ms":>10} {"step ms":>8} {"weak eff":>9}')
        for p in (2, 4, 8):
            for direct in (False, True):
                tag = 'direct RS+AG' if direct else 'ring (1 link)'
                # edge blocks: all-reduce S1, S2 forward and backward; arena: tail under layer-1 backward, prefix exposed
                coll = 2 * (t_allreduce(S1, p, bw, alpha, direct) + t_allreduce(S2, p, bw, alpha, direct))
                tail, prefix = t_allreduce(other, p, bw, alpha, direct), t_allreduce(table, p, bw, alpha, direct)
                # hidden: each layer exchange overlaps its self-loop GEMM / loop-gradient work (~30 us each at c2, ~100 us at c4);
                # the arena tail overlaps layer 1's backward entirely
                per_overlap = 30e-6 if c['h'] == 200 else 100e-6
                hidden = min(coll, 4 * per_overlap) + tail
                step = c['t1_ms'] * 1e-3 + coll + tail + prefix - hidden
                print(f'{p:5d} {"edge / " + tag:>22} {(coll + tail + prefix) * 1e3:14.3f} {hidden * 1e3:10.3f} {step * 1e3:8.3f} '
                      f'{c["t1_ms"] * 1e-3 / step:9.2f}')
                # destination rows: all-gather S1, S2 forward, reduce-scatter backward; node-level work is split over ranks too
                coll_r = 2 * (t_gather(S1, p, bw, alpha, direct) + t_gather(S2, p, bw, alpha, direct))
                arena_r = t_allreduce(table + other, p, bw, alpha, direct)
                step_r = c['t1_ms'] * 1e-3 * ((1 - c['node_share']) + c['node_share'] / p) + coll_r + arena_r
                print(f'{p:5d} {"rows / " + tag:>22} {(coll_r + arena_r) * 1e3:14.3f} {0.0:10.3f} {step_r * 1e3:8.3f} '
                      f'{c["t1_ms"] * 1e-3 / step_r:9.2f}')


# ---- STRONG scaling (bench.py's default with more than one rank: ONE graph, BASELINE configs[3..4]) ------------------------------
STRONG = {
    # measured 1-GPU steps (profiles/round5); node_share = the part of the step that is row-wise over nodes / triplets / parameters
    # (self-loop products, epilogues, reparameterisation, KL, MMD, decoder, clip + Adam): replicated under edge sharding, divided
    # under the row partition; the rest is K1 (aggregations + grad-W), divided by both.  k1_rel_gain: what a relation shard buys K1
    # itself -- a rank then touches R/p relation types, so the phase kernels stage 1/p of the table per tile (FB15k-237 at h = 500:
    # 4 phases instead of 32 -> the measured no-barrier / no-staging bound, profiles/round5/phase_ablation.txt; 1 M nodes: the
    # 1.3x weight re-fetch traffic goes) -- applied to the K1 share only
    'c2 (FB15k-237, h=200)': dict(N=14541, h=200, other_mb=2.9, t1_ms=1.005, node_share=0.45, k1_rel_gain=1.0),
    'c4 (FB15k-237, h=500)': dict(N=14541, h=500, other_mb=14.6, t1_ms=3.00, node_share=0.38, k1_rel_gain=1.35),
    'c5 (1 M entities, 50 M edges, h=200)': dict(N=1_000_000, h=200, other_mb=3.0, t1_ms=74.9, node_share=0.22, k1_rel_gain=1.15),
}


def strong(bw, alpha):
    """Predicted STRONG-scaling step time and speed-up t(1) / t(p).
      edge          edge blocks by relation + all-reduce of node rows (north_star as written): node-level work replicated
      edge+shard    ... with the optimiser sharded (reduce-scatter of the gradient arena, clip + Adam on 1/p of it, all-gather of
                    the updated parameters)
      rel+rows      north_star's relation shard WITHOUT the replicated node-level work: K1 over the rank's relations produces partial
                    rows of the whole table, REDUCE-SCATTER gives every rank the rows it owns, the node-level work runs on N/p rows,
                    ALL-GATHER rebuilds the table for the next layer (backward: all-gather of dL/dh, K1^T, reduce-scatter) -- twice
                    the bytes of the row partition per layer, K1 itself faster (k1_rel_gain).  Modelled, not built: see the table
      rows          destination-row partition (all-gather forward, reduce-scatter backward, node-level work divided): what is built
    Collectives 'direct' (every peer on its own link) / 'ring' (one link).  `hidden` = the fraction of every layer exchange that
    runs under kernels: 0.5 is what the 2-rank shared-GPU runs showed; 1.0 is the bound no schedule can beat (the arena exchange
    of the optimiser is never hidden: it sits between backward and the update)."""
    for name, c in STRONG.items():
        S1, S2 = c['N'] * c['h'] * 4, c['N'] * 2 * c['h'] * 4
        arena = S1 + c['other_mb'] * 1e6
        t1 = c['t1_ms'] * 1e-3
        adam = min(0.3 * t1 * c['node_share'], 4 * arena * 2 / 4.0e12)      # clip + Adam: ~8 arena passes at ~4 TB/s
        print(f'\n{name}: 1-GPU step {c["t1_ms"]} ms, node-level share {c["node_share"]}, S1 {S1 / 1e6:.1f} MB, S2 {S2 / 1e6:.1f} MB, '
              f'arena {arena / 1e6:.1f} MB (STRONG scaling: one graph)')
        print(f'{"ranks":>5} {"scheme":>34} {"compute ms":>11} {"layer exchanges ms":>19} {"arena ms":>9} {"step ms":>22} '
              f'{"speed-up":>9} {"all hidden":>11}')
        for p in (2, 4, 8):
            for direct in (True, False):
                tag = 'direct' if direct else 'ring'
                k1, node = t1 * (1 - c['node_share']), t1 * c['node_share']
                ar = 2 * (t_allreduce(S1, p, bw, alpha, direct) + t_allreduce(S2, p, bw, alpha, direct))
                # rows: four exchanges of S1 per step -- all-gather of h1 (layer 1 -> 2) and of z (-> decoder) forward, reduce-scatter of
                # dL/dz and dL/dh1 backward (h2 = [mean | pre-variance] stays on its owner: the reparameterisation is row-wise)
                ag = 4 * t_gather(S1, p, bw, alpha, direct)
                # ... pipelined (GV_DIST_ROW_CHUNKS = C, built): the h1 pair leaves in C blocks under the layers' own aggregations, 1/C
                # of it stays exposed; the z pair has only the rank's KL pass beside it
                C = 4
                one = t_gather(S1, p, bw, alpha, direct)
                # what stays exposed of the pipelined h1 pair: the last block's transfer + one latency per block; the z pair whole
                pipe_exposed = 2 * one + 2 * ((one - alpha) / C + C * alpha)
                # rel+rows: layer 1 forward RS(S1) [its input is the replicated table], layer 2 forward AG(S1) + RS(S2), AG(S1) of z; backward
                # RS(S1) of dL/dz, AG(S2) + RS(S1), AG(S1): six exchanges of S1 and two of S2 per step
                relrows = 6 * t_gather(S1, p, bw, alpha, direct) + 2 * t_gather(S2, p, bw, alpha, direct)
                rows = [('edge / ' + tag, k1 / p + node, ar, t_allreduce(arena, p, bw, alpha, direct)),
                        ('edge+sharded Adam / ' + tag, k1 / p + node - adam * (1 - 1 / p), ar, 2 * t_gather(arena, p, bw, alpha, direct)),
                        ('rel+rows, sharded Adam / ' + tag, k1 / p / c['k1_rel_gain'] + node / p, relrows, 2 * t_gather(arena, p, bw, alpha, direct)),
                        ('rows / ' + tag, (k1 + node) / p, ag, t_allreduce(arena, p, bw, alpha, direct)),
                        ('rows, h1 pair in 4 blocks / ' + tag, (k1 + node) / p, pipe_exposed, t_allreduce(arena, p, bw, alpha, direct)),
                        ('rows+sharded Adam / ' + tag, (k1 + node) / p, ag, 2 * t_gather(arena, p, bw, alpha, direct))]
                for label, comp, layer, ar_t in rows:
                    # exposed share of the layer exchanges: the edge scheme's all-reduces run in two destination-row blocks under the next block's
                    # aggregation and the self-loop product (half, as the 2-rank shared-GPU runs showed); the row scheme's single
                    # all-gather / reduce-scatter per pair has only the rank's self-loop product beside it (counted whole); 'in 4 blocks':
                    # `layer` already is what stays exposed
                    step = comp + (0.5 if label.startswith('edge') or label.startswith('rel') else 1.0) * layer + ar_t
                    best = comp + ar_t                       # every layer exchange hidden
                    print(f'{p:5d} {label:>34} {comp * 1e3:11.3f} {layer * 1e3:19.3f} {ar_t * 1e3:9.3f} {step * 1e3:22.3f} '
                          f'{t1 / step:9.2f} {t1 / best:11.2f}')


if __name__ == '__main__':
    main()
    import sys
    _a = argparse.ArgumentParser()
    _a.add_argument('--eff', type=float, default=0.8)
    _a.add_argument('--alpha-us', type=float, default=12.0)
    _o = _a.parse_args()
    print('\n================ STRONG scaling ================')
    strong(153e9 * _o.eff, _o.alpha_us * 1e-6)
