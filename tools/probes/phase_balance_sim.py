"""Host-side model of the relation-phase K1 launches' barrier cost on the FB15k-237-shaped graph (no GPU): a phase of a tile ends with the
wave that holds the longest list, so the launch's list time is ~ the SUM over tiles and phases of the longest per-wave list.  Prints
that sum against its mean (perfect balance = 1.00x) for the snake deal of items to waves and for the greedy per-phase choice
(indices.PhaseOrder.build, GV_PHASE_GREEDY), for several piece lengths of hub rows.
    python tools/probes/phase_balance_sim.py [dst|src]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib, sampling  # noqa: E402
from gcn_vae_amd.data import FB15K237, synthetic_kg  # noqa: E402

cfg = FB15K237
data = synthetic_kg(cfg['num_nodes'], cfg['num_rels'], cfg['n_train'], seed=0)
g, rel, _ = sampling.build_test_graph(data.num_nodes, data.num_rels, data.train)
src0, dst0 = (t.numpy() for t in g.edges())
N, E, R = data.num_nodes, src0.size, 2 * data.num_rels
side = sys.argv[1] if len(sys.argv) > 1 else 'dst'


def run(P, Q, trans, label):
    src, dst = (dst0, src0) if trans else (src0, dst0)
    plan = (ctypes.c_int32 * 7)()
    lib.load().gv_rgcn_bdd_phase_plan(100, P, Q, trans, R, 160 * 1024, 1, 0, 0, ctypes.addressof(plan))
    _, parts, K, G, n_phases, _, threads = [int(v) for v in plan]
    nw = threads // 64
    T = nw * K
    order = np.lexsort((rel, src, dst))
    dst_s, rel_s = dst[order], rel[order]
    deg = np.bincount(dst_s, minlength=N)
    rowptr = np.concatenate([[0], np.cumsum(deg)])
    phase_of_edge = rel_s // G
    for chunk in (128, 64, 32):
        nch = np.maximum(1, -(-deg // chunk))
        rows = np.repeat(np.arange(N), nch)
        kin = np.arange(rows.size) - np.repeat(np.cumsum(nch) - nch, nch)
        beg = rowptr[rows] + kin * chunk
        end = np.minimum(beg + chunk, rowptr[rows + 1])
        n_items, size = rows.size, end - beg
        n_tiles = -(-n_items // T)
        item_of_edge = np.searchsorted(beg, np.arange(E), side='right') - 1
        H = np.zeros((n_items, n_phases), np.int32)
        np.add.at(H, (item_of_edge, phase_of_edge), 1)
        by_size = np.argsort(-size, kind='stable')
        j = np.arange(n_items)
        rnd, pos = j // n_tiles, j % n_tiles
        tile_s = np.where(rnd % 2 == 0, pos, n_tiles - 1 - pos)
        rw, pw = rnd // nw, rnd % nw
        wave_s = np.where(rw % 2 == 0, pw, nw - 1 - pw)
        tile_of, wave_of = np.empty(n_items, int), np.empty(n_items, int)
        tile_of[by_size], wave_of[by_size] = tile_s, wave_s

        def cost(wave):
            L = np.zeros((n_tiles, nw, n_phases), np.int64)
            np.add.at(L, (tile_of, wave), H)
            return L.max(1).sum()
        wave_g = np.empty(n_items, int)
        for t in range(n_tiles):
            idx = np.nonzero(tile_of == t)[0]
            idx = idx[np.argsort(-size[idx], kind='stable')]
            L, cnt = np.zeros((nw, n_phases), np.int64), np.zeros(nw, int)
            for it in idx:
                cand = L + H[it][None, :]
                c = np.maximum(L.max(0)[None, :], cand).sum(1).astype(float) + 1e-3 * cand.sum(1)
                c[cnt >= K] = 1e18
                w = int(np.argmin(c))
                L[w] += H[it]
                cnt[w] += 1
                wave_g[it] = w
        Lt = np.zeros((n_tiles, n_phases), np.int64)
        np.add.at(Lt, tile_of, H)
        Mx = np.zeros((n_tiles, n_phases), np.int64)
        np.maximum.at(Mx, tile_of, H)
        bound = np.maximum(-(-Lt // nw), Mx).sum()
        mean = E / nw
        print(f'{label} (K={K}, {n_phases} phases of {G} relations, {nw} waves) pieces of <= {chunk:3d}: {n_items} items, {n_tiles} tiles x {parts} parts | '
              f'sum of the longest lists: snake {cost(wave_of) / mean:.2f}x, greedy {cost(wave_g) / mean:.2f}x, bound for this tile membership {bound / mean:.2f}x',
              flush=True)


if side == 'dst':
    run(5, 10, 0, '5x10 forward')
    run(5, 5, 0, '5x5 forward')
else:
    run(10, 5, 1, '10x5 backward-x')
    run(5, 5, 1, '5x5 backward-x')
