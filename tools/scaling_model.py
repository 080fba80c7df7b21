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


if __name__ == '__main__':
    main()
