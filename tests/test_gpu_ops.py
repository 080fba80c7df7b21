"""HIP kernels (through the C ABI) against the CPU oracle, op by op.   pytest -m gpu

Tolerance: north_star asks for 1e-4 (fp32) against the reference's CPU path; every comparison
below uses rtol=1e-4 with an absolute floor of 1e-5 x the output scale."""
import numpy as np
import pytest
import torch

from oracle import kgvae as okg
from oracle import prob as oprob
from oracle import rgcn as orgcn

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol_scale=1e-5, msg=''):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    atol = atol_scale * max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: f'{msg}: {m}')


@pytest.fixture(scope='module')
def ops():
    from gcn_vae_amd import ops as _ops
    return _ops


def zipf_graph(n, e, r, seed, skew=1.1):
    rs = np.random.RandomState(seed)
    p = (np.arange(n) + 1.0) ** (-skew)
    p /= p.sum()
    src = rs.choice(n, size=e, p=p)
    dst = rs.choice(n, size=e, p=p)
    et = rs.randint(0, r, size=e)
    order = np.lexsort((et, src, dst))
    src, dst, et = src[order], dst[order], et[order]
    deg = np.bincount(dst, minlength=n).astype(np.float32)
    norm = (1.0 / np.maximum(deg, 1))[dst].astype(np.float32)
    return (torch.from_numpy(src), torch.from_numpy(dst), torch.from_numpy(et), torch.from_numpy(norm).view(-1, 1))


def test_segment_items(ops):
    rs = np.random.RandomState(0)
    deg = rs.randint(0, 40, size=200)
    deg[7], deg[50], deg[199] = 1000, 0, 257
    rowptr = np.zeros(201, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    for chunk in (16, 64, 256):
        seg = ops.build_segment_items(torch.from_numpy(rowptr).cuda(), chunk)
        items = seg.items[:seg.n_items].cpu().numpy()
        fix = seg.fix[:seg.n_fix].cpu().numpy()
        exp_items, exp_fix, slot = [], [], 0
        for s in range(200):
            nch = max(1, -(-deg[s] // chunk))
            if nch > 1:
                exp_fix.append((s, slot, nch, 0))
            for k in range(nch):
                b = rowptr[s] + k * chunk
                exp_items.append((s, b, min(rowptr[s + 1], b + chunk), slot + k if nch > 1 else -1))
            if nch > 1:
                slot += nch
        assert np.array_equal(items, np.array(exp_items, dtype=np.int32))
        assert np.array_equal(fix, np.array(exp_fix, dtype=np.int32).reshape(-1, 4))
        assert seg.n_slots == slot
        # sync-free form: the same entries followed by -1 padding, sized by the upper bounds
        ub = ops.build_segment_items(torch.from_numpy(rowptr).cuda(), chunk, n_edges=int(rowptr[-1]))
        ui, uf = ub.items[:ub.n_items].cpu().numpy(), ub.fix[:ub.n_fix].cpu().numpy()
        assert ub.n_items >= seg.n_items and ub.n_fix >= seg.n_fix and ub.n_slots >= seg.n_slots
        assert np.array_equal(ui[:seg.n_items], items) and (ui[seg.n_items:] == -1).all()
        assert np.array_equal(uf[:seg.n_fix], fix) and (uf[seg.n_fix:] == -1).all()


def test_sync_free_index_gives_identical_results(ops):
    """Upper-bound-sized work-item lists (no host synchronisation per batch) vs exact lists: same bits, incl. hub rows."""
    rs = np.random.RandomState(1)
    n, e, r, nb = 300, 9000, 12, 10
    dst = np.sort(np.minimum((rs.pareto(1.2, e) * 3).astype(np.int64), n - 1))      # a few hub rows > chunk
    src = rs.randint(0, n, e)
    et = torch.from_numpy(rs.randint(0, r, e)).cuda()
    x = torch.randn(n, 40, device='cuda')
    w = torch.randn(r, nb * 4 * 4, device='cuda')
    g = torch.randn(n, 40, device='cuda')
    coef = torch.rand(e, device='cuda')
    outs = []
    for sf in (False, True):
        gi = ops.GraphIndex(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n, chunk=64, dst_sorted=True if sf else None,
                            sync_free=sf)
        ri = ops.RelationIndex(gi, et, r, chunk=32)
        fwd = ops.bdd_aggregate(gi.by_dst.seg, gi.nbr_by_dst, ri.et_by_dst, coef, gi.by_dst.perm, x, w, nb, 4, 4)
        bwd = ops.bdd_aggregate(gi.by_src.seg, gi.nbr_by_src, ri.et_by_src, coef, gi.by_src.perm, g, w, nb, 4, 4, True)
        gw = ops.bdd_grad_weight(ri.by_rel.seg, ri.src_by_rel, ri.dst_by_rel, coef, ri.by_rel.perm, x, g, nb, 4, 4)
        outs.append((fwd, bwd, gw))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


PHASE_CASES = [(200, 200, 100), (200, 400, 100), (500, 500, 100), (500, 1000, 100), (20, 40, 10), (40, 40, 20)]


@pytest.mark.parametrize('fin,fout,nb', PHASE_CASES)
@pytest.mark.parametrize('lds,threads,rows,chunk,nbuf', [(163840, 1024, 0, 256, 2), (24576, 256, 4, 8, 2), (40960, 128, 0, 32, 1)])
def test_rel_graph_conv_relation_phases(ops, monkeypatch, fin, fout, nb, lds, threads, rows, chunk, nbuf):
    """K1 by relation phases (csrc/k_phase.hip: weights staged through LDS once per tile, K rows per wave in registers)
    against the oracle, forward and every gradient: the planned geometry; short LDS budgets (many phases, a last phase
    with fewer relations), small workgroups, 4 rows per wave, and hub rows split into 8-edge items."""
    si, so = fin // nb, fout // nb
    if rows == 4 and not (si == 2 and so == 2):
        rows = 0
    monkeypatch.setattr(ops.indices, 'K1_PHASES', '1')
    monkeypatch.setattr(ops.indices, 'PHASE_LDS_BYTES', lds)
    monkeypatch.setattr(ops.indices, 'PHASE_THREADS', threads)
    monkeypatch.setattr(ops.indices, 'PHASE_ROWS', rows)
    monkeypatch.setattr(ops.indices, 'PHASE_BUFFERS', nbuf)
    n, e, r = 300, 4000, 120
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + fout + nb)
    gen = torch.Generator().manual_seed(fin * 7 + nb)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8)
    gout = torch.randn(n, fout, generator=gen)
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=chunk)
    ridx = ops.RelationIndex(gidx, et.cuda(), r)
    ph = ridx.phase_order(gidx, 'dst', nb, si, so)
    assert ph is not None and ridx.phase_order(gidx, 'src', nb, so, si) is not None
    if chunk == 8:
        assert ph.n_fix > 0 and ph.n_tiles > 4 and ph.n_phases > 2
    for act_id, act in ((1, torch.relu), (0, None)):
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, act, dropout_keep=keep, dropout_p=0.2)
        ho.backward(gout)
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb,
                                    act_id, keep.cuda(), 1.0 / 0.8)
        hg.backward(gout.cuda())
        close(hg, ho, msg='forward')
        close(xg.grad, xo.grad, msg='grad_x')
        close(pg['weight'].grad, po['weight'].grad, msg='grad_weight')
        close(pg['loop_weight'].grad, po['loop_weight'].grad, msg='grad_loop')


def test_relation_phase_lists_cover_every_edge_once(ops, monkeypatch):
    """Index property: the (tile, phase, wave) lists partition the edges; every edge sits in the list of its relation's
    phase and of the wave that owns its row's item, and carries that item's slot; every item has exactly one slot."""
    monkeypatch.setattr(ops.indices, 'PHASE_LDS_BYTES', 16384)
    monkeypatch.setattr(ops.indices, 'PHASE_THREADS', 256)
    n, e, r, nb = 500, 3000, 37, 10
    src, dst, et, _ = zipf_graph(n, e, r, seed=3)
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=16)
    ridx = ops.RelationIndex(gidx, et.cuda(), r)
    for side, rows_of, blk in (('dst', dst, (2, 4)), ('src', src, (4, 2))):
        ph = ridx.phase_order(gidx, side, nb, *blk)
        nw, K, G = ph.threads // 64, ph.rows_per_wave, ph.rels_per_phase
        off, meta, perm = ph.off.cpu().numpy(), ph.meta.cpu().numpy(), ph.perm.cpu().numpy()
        ti = ph.tile_items.cpu().numpy()
        assert off[0] == 0 and off[-1] == e and (np.diff(off) >= 0).all() and len(off) == ph.n_tiles * nw * ph.n_phases + 1
        assert sorted(perm.tolist()) == list(range(e))
        seg = (gidx.by_dst if side == 'dst' else gidx.by_src).seg
        rows_with_slot = ti[..., 0][ti[..., 0] >= 0]
        assert len(rows_with_slot) == seg.n_items
        for t in range(ph.n_tiles):
            for p_ in range(ph.n_phases):
                for w in range(nw):
                    idx = (t * nw + w) * ph.n_phases + p_
                    assert (np.diff(meta[off[idx]:off[idx + 1]] & 15) >= 0).all()       # sorted by item slot
                    for pos in range(off[idx], off[idx + 1]):
                        k, rl = meta[pos] & 15, meta[pos] >> 4
                        assert k < K and rl < G
                        assert et.numpy()[perm[pos]] == p_ * G + rl
                        assert ti[t, w * K + k, 0] == rows_of.numpy()[perm[pos]]


CASES = [  # (in, out, num_bases)  -> block sizes; covers the fast instantiations and the generic kernel
    (200, 200, 100), (200, 400, 100),   # C2 layer 1 / 2   (2x2, 2x4)
    (16, 16, 4), (16, 32, 4),           # C1               (4x4, 4x8)
    (40, 40, 40), (40, 80, 40),         # 1x1, 1x2
    (30, 30, 6), (30, 60, 6),           # 5x5, 5x10 -> generic
    (200, 200, 20), (200, 400, 20),     # C3 (configs[2]): 10x10, 10x20 -> column-split kernels (k_agg_split / k_gradw_split)
    (100, 200, 10), (60, 60, 6),        # the same blocks with fewer of them (partly filled waves)
    (24, 12, 3),                        # 8x4 -> generic (non-transposed), bwd-x 4x8
    (500, 500, 100), (500, 1000, 100),  # C4 / the reference's default --n-hidden 500: 5x5, 5x10, two column parts
]


@pytest.mark.parametrize('fin,fout,nb', CASES)
@pytest.mark.parametrize('chunk', [8, 256])
def test_rel_graph_conv_fwd_bwd(ops, fin, fout, nb, chunk):
    n, e, r = 300, 4000, 120
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + fout + nb)
    gen = torch.Generator().manual_seed(fin * 7 + nb)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8)
    gout = torch.randn(n, fout, generator=gen)
    for act_id, act in ((1, torch.relu), (0, None)):
        # oracle
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, act, dropout_keep=keep, dropout_p=0.2)
        ho.backward(gout)
        # HIP
        gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=chunk)
        etc = et.cuda()
        ridx = ops.RelationIndex(gidx, etc, r, chunk=max(4, chunk // 2))
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb,
                                    act_id, keep.cuda(), 1.0 / 0.8)
        hg.backward(gout.cuda())
        close(hg, ho, msg='forward')
        close(xg.grad, xo.grad, msg='grad_x')
        close(pg['weight'].grad, po['weight'].grad, msg='grad_weight')
        close(pg['h_bias'].grad, po['h_bias'].grad, msg='grad_bias')
        close(pg['loop_weight'].grad, po['loop_weight'].grad, msg='grad_loop')


@pytest.mark.parametrize('fin,fout,nb,r', [(200, 200, 20, 22), (200, 400, 20, 22), (200, 400, 20, 20), (100, 100, 10, 13)])
@pytest.mark.parametrize('max_edges,shuffle', [(64, False), (8, False), (64, True)])
def test_rel_graph_conv_lds_resident_weights(ops, monkeypatch, fin, fout, nb, r, max_edges, shuffle):
    """K1 with all relation weights resident in LDS (csrc/k_lds.hip; BASELINE configs[2]'s shape: 22 directed relation types,
    20 blocks of 10x10 / 10x20) against the oracle, forward and every gradient: super-items of up to 64 edges (runs of short
    rows, 64-edge slices of the hub rows through the fix-up pass) and of up to 8 (many slices, many one-row items), edges in
    arbitrary order (coefficients read through the permutation), rows without edges (the launch's second loop)."""
    n, e = 400, 5000
    assert ops.lds_plan(r, nb, fin // nb, fout // nb) is not None and ops.lds_plan(r, nb, fout // nb, fin // nb) is not None
    assert ops.lds_plan(474, nb, fin // nb, fout // nb) is None          # the table must fit a CU's LDS
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + fout + nb + r)
    dst = torch.where(dst >= n - 7, torch.zeros_like(dst), dst)          # the last rows: in-degree 0
    if shuffle:
        perm = torch.randperm(e, generator=torch.Generator().manual_seed(2))
        src, dst, et, norm = src[perm], dst[perm], et[perm], norm[perm]
    else:
        order = np.lexsort((et.numpy(), src.numpy(), dst.numpy()))
        src, dst, et, norm = src[order], dst[order], et[order], norm[order]
    gen = torch.Generator().manual_seed(fin * 7 + nb)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8)
    gout = torch.randn(n, fout, generator=gen)
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n)
    assert (gidx.by_dst.perm is not None) == shuffle
    ridx = ops.RelationIndex(gidx, et.cuda(), r)
    real_plan = ops.lds_plan
    monkeypatch.setattr(ops, 'lds_plan', lambda *a, **k: None if real_plan(*a, **k) is None else
                        real_plan(*a, **k)[:2] + (max_edges,) + real_plan(*a, **k)[3:])
    od = gidx.lds_order('dst', max_edges)
    si = od.sitems[:od.n_sitems]
    assert od.n_fix > 0 and od.n_empty >= 7 and int((si[:, 1] - si[:, 0]).max()) <= max_edges
    assert int((si[:, 1] - si[:, 0]).sum()) == e and od.n_slots == int(od.fix[:od.n_fix, 2].sum())
    calls = []
    real = ops.bdd_aggregate_lds
    monkeypatch.setattr(ops, 'bdd_aggregate_lds', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for act_id, act in ((1, torch.relu), (0, None)):
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, act, dropout_keep=keep, dropout_p=0.2)
        ho.backward(gout)
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb,
                                    act_id, keep.cuda(), 1.0 / 0.8)
        hg.backward(gout.cuda())
        close(hg, ho, msg='forward')
        close(xg.grad, xo.grad, msg='grad_x')
        close(pg['weight'].grad, po['weight'].grad, msg='grad_weight')
        close(pg['h_bias'].grad, po['h_bias'].grad, msg='grad_bias')
        close(pg['loop_weight'].grad, po['loop_weight'].grad, msg='grad_loop')
    assert len(calls) == 4                                               # forward and backward-x of both runs took the LDS kernel
    # bitwise reproducible: a second run gives the same bits
    xg2 = x.cuda().requires_grad_(True)
    hg2 = ops.rel_graph_conv_bdd(xg2, pg['weight'].detach(), pg['h_bias'].detach(), pg['loop_weight'].detach(), norm.cuda(), gidx, ridx,
                                 nb, 0, keep.cuda(), 1.0 / 0.8)
    assert torch.equal(hg2, hg.detach())
    # no self loop, no bias, no norm, no dropout mask: absent epilogue operands
    ho3 = orgcn.rel_graph_conv(x, src, dst, et, None, {'weight': p['weight']}, 'bdd', nb, None)
    hg3 = ops.rel_graph_conv_bdd(x.cuda(), p['weight'].cuda(), None, None, None, gidx, ridx, nb, 0)
    close(hg3, ho3)
    assert float(hg3[n - 7:].abs().max()) == 0.0


@pytest.mark.parametrize('fin,fout,nb,r,chunk', [(200, 200, 100, 100, 256), (200, 400, 100, 100, 64), (400, 200, 100, 100, 16),
                                                 (200, 200, 200, 200, 256), (40, 80, 10, 10, 8)])        # (DGL clamps num_bases to num_rels)
def test_rel_graph_conv_rows_sorted_by_relation_reuse_the_weights_of_a_run(ops, monkeypatch, fin, fout, nb, r, chunk):
    """Static graphs hand the per-row kernels every row's edges sorted by relation (ops.RelationIndex.rel_sorted); the kernels keep a
    relation's weights in registers while consecutive edges share it (csrc/k_bdd.hip: r_keep / wkeep).  Hub rows (Zipf 1.1: a quarter of the edges in the top row) make the runs long: they cross the kernels' edge batches, the 64-edge metadata chunks and (small work items) item
    boundaries.  Against the oracle, forward and every gradient; and against the same launch without the re-ordering."""
    n, e = 300, 12000
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + fout + nb + r)
    order = np.lexsort((et.numpy(), src.numpy(), dst.numpy()))
    src, dst, et, norm = src[order], dst[order], et[order], norm[order]
    gen = torch.Generator().manual_seed(fin * 3 + nb)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    gout = torch.randn(n, fout, generator=gen)
    monkeypatch.setattr(ops, 'lds_plan', lambda *a, **k: None)                 # keep these shapes on the per-row kernels
    res = {}
    for runs in (True, False):
        monkeypatch.setattr(ops, 'K1_REL_RUNS', runs)
        gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=chunk)
        ridx = ops.RelationIndex(gidx, et.cuda(), r)
        if runs:
            nbr_r, et_r, eid_r, coef_r = ridx.rel_sorted(gidx, 'dst', norm.cuda())
            rp = gidx.by_dst.seg.rowptr.long()
            rows = torch.repeat_interleave(torch.arange(n, device='cuda'), rp[1:] - rp[:-1])
            key = rows * r + et_r.long()
            assert bool((key[1:] >= key[:-1]).all())                             # rows in place, relations ascending inside a row
            assert torch.equal(torch.sort(eid_r.long())[0], torch.arange(e, device='cuda'))
            assert torch.equal(coef_r.view(-1), norm.cuda().view(-1)[eid_r.long()]) and torch.equal(nbr_r.long(), src.cuda()[eid_r.long()])
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        hg = ops.rel_graph_conv_bdd(xg, pg['weight'], None, pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 1)
        hg.backward(gout.cuda())
        res[runs] = (hg.detach(), xg.grad, pg['weight'].grad, pg['loop_weight'].grad)
    xo = x.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, {k: v for k, v in po.items() if k != 'h_bias'}, 'bdd', nb, torch.relu)
    ho.backward(gout)
    want = (ho.detach(), xo.grad, po['weight'].grad, po['loop_weight'].grad)
    for name, a, b, c in zip(('forward', 'grad_x', 'grad_weight', 'grad_loop'), res[True], res[False], want):
        close(a, c, msg=name + ' (relation-sorted rows) vs oracle')
        close(a, b, rtol=1e-5, atol_scale=1e-6, msg=name + ': relation-sorted vs neighbour-sorted rows (summation order only)')


@pytest.mark.parametrize('fin,fout,nb,r', [(200, 200, 20, 22), (200, 400, 20, 22), (100, 200, 10, 22)])
def test_rel_graph_conv_lds_resident_bf16_operands(ops, fin, fout, nb, r):
    """BASELINE configs[2]'s precision on K1 (gv_rgcn_bdd_aggregate_lds with bf16_operands): the relation weights and the
    coefficient-scaled inputs rounded to bf16 (nearest even), products and sums in fp32 -- against the oracle's statement of
    exactly that (oracle/bf16.py, ``enabled(k1=True)``), forward, backward-x (bf16 operands too) and the fp32 weight gradient.
    The roundings are the same on both sides, so the tolerance stays that of a different summation order."""
    from oracle import bf16 as obf16
    n, e = 500, 6000
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + nb + r)
    dst = torch.where(dst >= n - 5, torch.zeros_like(dst), dst)
    order = np.lexsort((et.numpy(), src.numpy(), dst.numpy()))
    src, dst, et, norm = src[order], dst[order], et[order], norm[order]
    gen = torch.Generator().manual_seed(fin + 3 * nb)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8)
    gout = torch.randn(n, fout, generator=gen)
    with ops.gemm_precision('bf16'):
        assert ops.lds_plan(r, nb, fin // nb, fout // nb)[3] and ops.lds_plan(r, nb, fout // nb, fin // nb)[3]
        fp, bp = ops.lds_plan(r, nb, fin // nb, fout // nb), ops.lds_plan(r, nb, fin // nb, fout // nb, bf=False)
        assert fp[0] <= bp[0]                                            # never more column parts than the fp32 table needs
        gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n)
        ridx = ops.RelationIndex(gidx, et.cuda(), r)
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 1, keep.cuda(),
                                    1.0 / 0.8)
        hg.backward(gout.cuda())
    with obf16.enabled(True, k1=True):
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, torch.relu, dropout_keep=keep, dropout_p=0.2)
        ho.backward(gout)
    close(hg, ho, rtol=2e-4, atol_scale=2e-5, msg='forward')
    close(xg.grad, xo.grad, rtol=2e-4, atol_scale=2e-5, msg='grad_x')
    close(pg['weight'].grad, po['weight'].grad, rtol=2e-4, atol_scale=2e-5, msg='grad_weight')
    # and the rounding is really there: the fp32 oracle differs by about a bf16 ulp of the operands
    ho32 = orgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', nb, torch.relu, dropout_keep=keep, dropout_p=0.2)
    assert float((hg.detach().cpu() - ho32).abs().max()) > 1e-4 * float(ho32.abs().max())


def test_rel_graph_conv_unsorted_edges_and_empty_rows(ops):
    n, e, r, fin, fout, nb = 120, 900, 5, 8, 16, 4
    src, dst, et, norm = zipf_graph(n, e, r, seed=3)
    perm = torch.randperm(e, generator=torch.Generator().manual_seed(1))
    src, dst, et, norm = src[perm], dst[perm], et[perm], norm[perm]
    dst = torch.where(dst >= n - 5, torch.zeros_like(dst), dst)      # last 5 nodes: in-degree 0
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    ho = orgcn.rel_graph_conv(x, src, dst, et, norm, p, 'bdd', nb, torch.relu)
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n)
    assert gidx.by_dst.perm is not None
    ridx = gidx.relation_index(et.cuda(), r)
    hg = ops.rel_graph_conv_bdd(x.cuda(), p['weight'].cuda(), p['h_bias'].cuda(), p['loop_weight'].cuda(),
                                norm.cuda(), gidx, ridx, nb, 1)
    close(hg, ho)
    # no self loop, no bias, no norm
    ho2 = orgcn.rel_graph_conv(x, src, dst, et, None, {'weight': p['weight']}, 'bdd', nb, None)
    hg2 = ops.rel_graph_conv_bdd(x.cuda(), p['weight'].cuda(), None, None, None, gidx, ridx, nb, 0)
    close(hg2, ho2)
    assert float(hg2[n - 5:].abs().max()) == 0.0


@pytest.mark.parametrize('m,n,k', [(300, 200, 200), (1000, 400, 200), (37, 19, 53), (128, 64, 16), (5, 3, 2),
                                   (200, 400, 3000), (260, 500, 1000), (64, 64, 40), (68, 132, 100), (4, 4, 4)])
def test_gemm_all_layouts(ops, m, n, k):
    gen = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=gen)
    b = torch.randn(k, n, generator=gen)
    bias = torch.randn(n, generator=gen)
    ref = a.double() @ b.double()
    close(ops.gemm(a.cuda(), b.cuda()), ref)
    close(ops.gemm(a.t().contiguous().cuda(), b.cuda(), trans_a=True), ref)
    close(ops.gemm(a.cuda(), b.t().contiguous().cuda(), trans_b=True), ref)
    close(ops.gemm(a.t().contiguous().cuda(), b.t().contiguous().cuda(), trans_a=True, trans_b=True), ref)
    close(ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), act=1), torch.relu(ref + bias.double()))
    c0 = torch.randn(m, n, generator=gen)
    close(ops.gemm(a.cuda(), b.cuda(), out=c0.clone().cuda(), accumulate=True), ref + c0.double())
    for sk in (2, 7):
        close(ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), split_k=sk), ref + bias.double())
        close(ops.gemm(a.t().contiguous().cuda(), b.cuda(), trans_a=True, split_k=sk), ref)
    close(ops.colsum(a.cuda()), a.double().sum(0))


def test_linear_fn(ops):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(257, 200, generator=gen)
    w = torch.randn(400, 200, generator=gen) * 0.1
    b = torch.randn(400, generator=gen) * 0.1
    go = torch.randn(257, 400, generator=gen)
    for act_id, f in ((1, torch.relu), (0, lambda t: t)):
        xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
        yo = f(torch.nn.functional.linear(xo, wo, bo))
        yo.backward(go)
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        yg = ops.linear(xg, wg, bg, act_id)
        yg.backward(go.cuda())
        close(yg, yo)
        close(xg.grad, xo.grad)
        close(wg.grad, wo.grad)
        close(bg.grad, bo.grad)


def test_embedding(ops):
    gen = torch.Generator().manual_seed(0)
    table = torch.randn(50, 24, generator=gen)
    ids = torch.tensor([3, 7, 7, 0, 49, 3, 3]).view(-1, 1)
    go = torch.randn(7, 24, generator=gen)
    to = table.clone().requires_grad_(True)
    torch.nn.functional.embedding(ids.squeeze(), to).backward(go)
    tg = table.cuda().requires_grad_(True)
    out = ops.embedding(tg, ids.cuda())
    out.backward(go.cuda())
    close(out, table[ids.squeeze()])
    close(tg.grad, to.grad)


def test_reparam(ops):
    gen = torch.Generator().manual_seed(1)
    h2 = torch.randn(301, 80, generator=gen) * 3
    h2[0, 40:48] = torch.tensor([-30.0, -5.0, 0.0, 5.0, 19.9, 20.1, 50.0, 1e-3])
    eps = torch.randn(301, 40, generator=gen)
    gz, gm, gv = (torch.randn(301, 40, generator=gen) for _ in range(3))
    ho = h2.clone().requires_grad_(True)
    m, v = oprob.gaussian_parameters(ho)
    z = oprob.sample_gaussian(m, v, eps)
    (z * gz + m * gm + v * gv).sum().backward()
    hg = h2.cuda().requires_grad_(True)
    zg, mg, vg = ops.reparam(hg, eps.cuda())
    (zg * gz.cuda() + mg * gm.cuda() + vg * gv.cuda()).sum().backward()
    close(zg, z)
    close(mg, m)
    close(vg, v)
    close(hg.grad, ho.grad)


@pytest.mark.parametrize('h', [200, 500, 36])      # 500: 1x1 blocks over several column parts; 36: generic sizes
def test_distmult_bce_and_score(ops, h):
    gen = torch.Generator().manual_seed(2)
    n, r, T = 400, 9, 5000
    emb = torch.randn(n, h, generator=gen) * 0.3
    w = torch.randn(r, h, generator=gen) * 0.3
    rs = np.random.RandomState(0)
    p = (np.arange(n) + 1.0) ** -1.0
    p /= p.sum()
    trip = torch.from_numpy(np.stack([rs.choice(n, T, p=p), rs.randint(0, r, T), rs.choice(n, T, p=p)], 1))
    labels = (torch.rand(T, generator=gen) > 0.7).float()
    flp = torch.tensor(-0.37)
    eo, wo, fo = emb.clone().requires_grad_(True), w.clone().requires_grad_(True), flp.clone().requires_grad_(True)
    so = okg.distmult_score(eo, wo, trip) + fo
    lo = torch.nn.functional.binary_cross_entropy_with_logits(so, labels)
    (lo * 1.7).backward()
    tidx = ops.TripletIndex(trip.cuda(), n, r, chunk=64, chunk_rel=32)
    eg, wg, fg = emb.cuda().requires_grad_(True), w.cuda().requires_grad_(True), flp.cuda().requires_grad_(True)
    lg, sg = ops.distmult_bce(eg, wg, fg, labels.cuda(), tidx)
    (lg * 1.7).backward()
    close(sg, so)
    close(lg, lo)
    close(eg.grad, eo.grad)
    close(wg.grad, wo.grad)
    close(fg.grad, fo.grad)
    # calc_score path with an arbitrary upstream gradient
    gs = torch.randn(T, generator=gen)
    eo2, wo2 = emb.clone().requires_grad_(True), w.clone().requires_grad_(True)
    okg.distmult_score(eo2, wo2, trip).backward(gs)
    eg2, wg2 = emb.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    s2 = ops.distmult_score(eg2, wg2, tidx)
    s2.backward(gs.cuda())
    close(s2, so - flp)
    close(eg2.grad, eo2.grad)
    close(wg2.grad, wo2.grad)


def test_mean_sq_and_kl(ops):
    gen = torch.Generator().manual_seed(3)
    n, h, k = 333, 200, 10
    z = torch.randn(n, h, generator=gen)
    m = torch.randn(n, h, generator=gen) * 0.5
    v = torch.rand(n, h, generator=gen) + 0.2
    z_pre = torch.randn(1, 2 * k, h, generator=gen) / np.sqrt(k * h) * 30
    flp = torch.tensor(0.83)
    xo = z.clone().requires_grad_(True)
    (xo.pow(2).mean() * 3).backward()
    xg = z.cuda().requires_grad_(True)
    ms = ops.mean_sq(xg)
    (ms * 3).backward()
    close(ms, z.pow(2).mean())
    close(xg.grad, xo.grad)
    to = [t.clone().requires_grad_(True) for t in (z, m, v, z_pre, flp)]
    klo = okg.kl_term(to[0], to[1], to[2], to[3], to[4])
    (klo * 2.5).backward()
    tg = [t.cuda().requires_grad_(True) for t in (z, m, v, z_pre, flp)]
    klg = ops.kl_to_mixture(tg[0], tg[1], tg[2], tg[3].squeeze(0), tg[4])
    (klg * 2.5).backward()
    close(klg, klo)
    for a, b, nm in zip(tg, to, ('z', 'm', 'v', 'z_pre', 'flp')):
        close(a.grad, b.grad, msg=nm)
    # flp = None reads as 0 (documented fix of the reference's None + tensor crash)
    close(ops.kl_to_mixture(tg[0].detach(), tg[1].detach(), tg[2].detach(), tg[3].detach().squeeze(0), None),
          okg.kl_term(z, m, v, z_pre, None))


@pytest.mark.parametrize('n,h,k', [(5, 16, 3), (1000, 256, 10), (77, 200, 16), (64, 8, 4), (2049, 200, 10), (300, 260, 10), (40, 30, 5),
                                   (120, 500, 10), (33, 1024, 4), (40001, 200, 10)])      # (> 16 384 nodes: a wave's second iteration)
def test_kl_kernel_forms_over_shapes(ops, n, h, k):
    """gv_kl_fwd / gv_kl_bwd over the shapes that select their kernel forms -- forward: four nodes per wave (h <= 512, k <= 16; node
    counts that leave rows of a wave without a node), a lane on four columns (wider rows), a lane per column (h % 4 != 0); backward: a
    lane on four columns in 256-column tiles (k <= 10, the table within 64 KB of LDS), a lane per column otherwise -- against the
    oracle: the KL term and every gradient."""
    gen = torch.Generator().manual_seed(n + h + k)
    z = torch.randn(n, h, generator=gen)
    m = torch.randn(n, h, generator=gen) * 0.5
    v = torch.rand(n, h, generator=gen) + 0.2
    z_pre = torch.randn(1, 2 * k, h, generator=gen) / np.sqrt(k * h) * 30
    flp = torch.tensor(-0.4)
    to = [t.clone().requires_grad_(True) for t in (z, m, v, z_pre, flp)]
    klo = okg.kl_term(*to)
    (klo * 1.5).backward()
    tg = [t.cuda().requires_grad_(True) for t in (z, m, v, z_pre, flp)]
    klg = ops.kl_to_mixture(tg[0], tg[1], tg[2], tg[3].squeeze(0), tg[4])
    (klg * 1.5).backward()
    close(klg, klo)
    for a, b, nm in zip(tg, to, ('z', 'm', 'v', 'z_pre', 'flp')):
        close(a.grad, b.grad, msg=nm)


def test_product_rejects_cpu_tensors(ops):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gemm(torch.randn(4, 4), torch.randn(4, 4))


def test_mmd_and_prior_sample(ops):
    from oracle import prob as op_
    gen = torch.Generator().manual_seed(4)
    k, h = 10, 200
    z_pre = torch.randn(1, 2 * k, h, generator=gen) * 0.3
    eps = torch.randn(200, h, generator=gen)
    y = torch.randn(150, h, generator=gen) * 0.8
    zo, yo = z_pre.clone().requires_grad_(True), y.clone().requires_grad_(True)
    m_mix, v_mix = op_.gaussian_parameters(zo, dim=1)
    xo = op_.sample_gaussian(m_mix, v_mix, eps, repeat=20)
    mo = okg.rbf_kernel(xo, xo).mean() + okg.rbf_kernel(yo, yo).mean() - 2 * okg.rbf_kernel(xo, yo).mean()
    (mo * 1.3).backward()
    zg, yg = z_pre.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    xg = ops.prior_sample(zg.squeeze(0), eps.cuda())
    mg = ops.mmd(xg, yg)
    (mg * 1.3).backward()
    close(xg, xo)
    close(mg, mo, atol_scale=1e-6)
    close(zg.grad, zo.grad, rtol=2e-4, atol_scale=2e-5)
    close(yg.grad, yo.grad, rtol=2e-4, atol_scale=2e-5)


@pytest.mark.parametrize('sx,sy,h', [(7, 130, 16), (129, 3, 256), (64, 64, 500), (50, 70, 30), (200, 200, 72), (1, 1, 8)])
def test_mmd_kernel_forms_over_shapes(ops, sx, sy, h):
    """gv_mmd_fwd / gv_mmd_bwd over the shapes that select their kernel forms -- a row of 16 lanes per partner row (h % 4 == 0:
    forward up to 512 columns, backward up to 256; set sizes that leave groups and whole iterations without a partner row), a lane per
    column otherwise -- against the oracle's kernel means: the value and both gradients."""
    gen = torch.Generator().manual_seed(sx * 7 + sy * 3 + h)
    x, y = torch.randn(sx, h, generator=gen), torch.randn(sy, h, generator=gen) * 0.7 + 0.1
    xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    mo = okg.rbf_kernel(xo, xo).mean() + okg.rbf_kernel(yo, yo).mean() - 2 * okg.rbf_kernel(xo, yo).mean()
    (mo * 0.7).backward()
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    mg = ops.mmd(xg, yg)
    (mg * 0.7).backward()
    close(mg, mo, atol_scale=1e-6)
    close(xg.grad, xo.grad, rtol=2e-4, atol_scale=2e-5)
    close(yg.grad, yo.grad, rtol=2e-4, atol_scale=2e-5)


def test_flat_adam_matches_torch_clip_and_adam(ops):
    from gcn_vae_amd.optim import FlatAdam
    gen = torch.Generator().manual_seed(6)
    shapes = [(37, 5), (200,), (3, 4, 5), (1,)]
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    opt_r = torch.optim.Adam(ref, lr=1e-2)
    opt_m = FlatAdam(mine, lr=1e-2, max_grad_norm=0.7)
    for it in range(5):
        grads = [torch.randn(*s, generator=gen) * (3.0 if it % 2 else 0.05) for s in shapes]
        for p, gr in zip(ref, grads):
            p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 0.7)
        opt_r.step()
        opt_m.zero_grad()
        for p, gr in zip(mine, grads):
            p.grad += gr.cuda()
        opt_m.step()
        for a, b in zip(mine, ref):
            close(a, b, rtol=2e-5, atol_scale=1e-6)


@pytest.mark.parametrize('fin,fout,nb', [(200, 200, 100), (200, 400, 100), (16, 16, 4), (16, 32, 4), (64, 64, 16)])
def test_lane_packed_weights_are_bit_identical(ops, fin, fout, nb):
    n, e, r = 300, 5000, 120
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + nb)
    gen = torch.Generator().manual_seed(3)
    si, so = fin // nb, fout // nb
    x = torch.randn(n, fin, generator=gen).cuda()
    g = torch.randn(n, fout, generator=gen).cuda()
    w = torch.randn(r, nb * si * so, generator=gen).cuda()
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=32)
    ridx = gidx.relation_index(et.cuda(), r)
    nrm = norm.cuda().reshape(-1)
    for (seg, nbr, ety, perm, feat, p, q, tr) in (
            (gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, gidx.by_dst.perm, x, si, so, False),
            (gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, gidx.by_src.perm, g, so, si, True)):
        assert ops.pack_supported(nb, p, q, tr)
        ref = ops.bdd_aggregate(seg, nbr, ety, nrm, perm, feat, w, nb, p, q, tr)
        got = ops.bdd_aggregate(seg, nbr, ety, nrm, perm, feat, ops.pack_weight(w, nb, p, q, tr), nb, p, q, tr, packed=True)
        assert torch.equal(ref, got)
    assert not ops.pack_supported(200, 1, 1, False) and not ops.pack_supported(20, 10, 10, False)


@pytest.mark.parametrize('fin,fout,nb,n,e,r', [(500, 1000, 100, 400, 9000, 120), (200, 400, 100, 300, 5000, 120), (16, 16, 4, 50, 0, 5),
                                               (40, 40, 40, 700, 30000, 47)])
def test_relation_grouped_order_equals_plain_order(ops, fin, fout, nb, n, e, r):
    """RelationIndex.grouped_order (a row's edges by relation range, one range per XCD -- for weight tables that overflow
    an XCD's L2): the layer's forward and every gradient equal the plain (neighbour, relation) order up to fp32 rounding."""
    src, dst, et, norm = zipf_graph(n, e, r, seed=fin + e)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8).cuda()
    gout = torch.randn(n, fout, generator=gen).cuda()
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=64)          # exact-size index (static graph)
    ridx = gidx.relation_index(et.cuda(), r)
    if e:
        seg, nbr, ety, perm = ridx.grouped_order(gidx, 'dst', chunk=64)
        it = seg.items[seg.items[:, 0] >= 0]
        assert int((it[:, 2] - it[:, 1]).sum()) == e and int((it[:, 2] - it[:, 1]).max()) <= 64
        assert torch.equal(torch.sort(perm)[0], torch.arange(e, device='cuda'))
        blocks = torch.arange(seg.items.shape[0], device='cuda') // 4 % 8          # the XCD a position is dealt to
        valid = seg.items[:, 0] >= 0
        cnt = torch.bincount(et.cuda(), minlength=r)
        grp_of_rel = torch.clamp((torch.cumsum(cnt, 0) - cnt) * 8 // e, max=7)
        nonempty = valid & (seg.items[:, 2] > seg.items[:, 1])
        first_edge_group = grp_of_rel[ety.long()[seg.items[nonempty, 1].long()]]
        assert torch.equal(first_edge_group, blocks[nonempty])                          # every item sits on its group's XCD
    res = {}
    old = ops.REL_GROUPS
    try:
        for mode in ('1', '0'):
            ops.REL_GROUPS = mode
            xg = x.cuda().requires_grad_(True)
            pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
            h = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 1, keep,
                                       1.25)
            h.backward(gout)
            res[mode] = [h.detach(), xg.grad] + [pg[k].grad for k in sorted(pg)]
    finally:
        ops.REL_GROUPS = old
    for a, b in zip(res['1'], res['0']):
        close(a, b, rtol=2e-5, atol_scale=2e-6, msg='grouped vs plain order')


def test_edge_shard_code_path_equals_fused_path(ops):
    """The multi-GPU branch of the layer (chunked raw aggregate -> reduce hook -> separate epilogue; reduce hook on the
    backward gradient) with a no-op hook must reproduce the single-GPU fused path."""
    class Done:
        def wait(self):
            return None
    calls = []

    def hook(t):
        calls.append(tuple(t.shape))
        return Done()
    n, e, r, fin, fout, nb = 500, 9000, 120, 200, 400, 100
    src, dst, et, norm = zipf_graph(n, e, r, seed=11)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    p['h_bias'] = torch.randn(fout, generator=gen) * 0.1
    keep = (torch.rand(n, fout, generator=gen) > 0.2).to(torch.uint8).cuda()
    gout = torch.randn(n, fout, generator=gen).cuda()
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n, chunk=64)
    ridx = gidx.relation_index(et.cuda(), r)
    res = []
    for hk in (None, hook):
        xg = x.cuda().requires_grad_(True)
        pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
        h = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 1, keep,
                                   1.25, hk)
        h.backward(gout)
        res.append((h, xg.grad, pg['weight'].grad, pg['h_bias'].grad, pg['loop_weight'].grad))
    for a, b in zip(*res):
        close(a, b, rtol=1e-5, atol_scale=1e-6)
    chunks = gidx.dst_chunks(ops.DIST_FWD_CHUNKS)
    assert chunks[0][0] == 0 and chunks[-1][1] == n and sum(c[2].n_items for c in chunks) == gidx.by_dst.seg.n_items
    assert len(calls) == len(chunks) + 1 and calls[-1] == (n, fout)          # forward row blocks + one backward gradient


def test_basis_regularizer_layer(ops):
    """RelGraphConv(regularizer='basis') -- W_r = sum_b w_comp[r,b] V_b through the f32 GEMM, then the
    generic aggregation kernels with one dense (in x out) block per relation; forward + all gradients vs the oracle."""
    from gcn_vae_amd.graph import KGraph
    from gcn_vae_amd.layers import RelGraphConv
    n, e, r, fin, fout, nbases = 90, 700, 12, 10, 14, 5
    src, dst, et, norm = zipf_graph(n, e, r, seed=8)
    torch.manual_seed(0)
    for nb in (nbases, r):          # nb < R: w_comp present; nb == R: plain per-relation matrices
        layer = RelGraphConv(fin, fout, r, 'basis', nb, activation=torch.tanh, self_loop=True, dropout=0.0)
        with torch.no_grad():
            layer.h_bias.normal_(0, 0.1)
        assert ('w_comp' in dict(layer.named_parameters())) == (nb < r)
        x = torch.randn(n, fin)
        gout = torch.randn(n, fout)
        po = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        xo = x.clone().requires_grad_(True)
        ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'basis', nb, torch.tanh)
        ho.backward(gout)
        g = KGraph()
        g.add_nodes(n)
        g.add_edges(src, dst)
        layer = layer.cuda()
        xg = x.cuda().requires_grad_(True)
        hg = layer(g, xg, et.cuda(), norm.cuda())
        hg.backward(gout.cuda())
        close(hg, ho, msg='basis forward')
        close(xg.grad, xo.grad, msg='basis grad_x')
        for k, v in layer.named_parameters():
            close(v.grad, po[k].grad, rtol=2e-4, atol_scale=2e-5, msg='basis grad ' + k)


@pytest.mark.parametrize('self_loop', [True, False])
def test_basis_layer_with_integer_id_features(ops, self_loop):
    """SURVEY 8(f-3), second half: the input layer of kgvae/entity_classify.py:25-34, :63 -- ``features = arange(num_nodes)``
    into ``RelGraphConv(num_nodes, h, R, "basis", num_bases, activation=relu, self_loop=...)``: a message is ROW
    (etype, id) of the relation's matrix (DGL bmm_maybe_select), the self-loop term a row of loop_weight.  Forward and all
    gradients (weight, w_comp, h_bias, loop_weight) against the oracle; a permuted id list as well as arange; the bdd
    regulariser refuses integer ids as DGL does."""
    from gcn_vae_amd.graph import KGraph
    from gcn_vae_amd.layers import RelGraphConv
    n, e, r, fout = 150, 1100, 10, 16
    src, dst, et, norm = zipf_graph(n, e, r, seed=21)
    torch.manual_seed(3)
    gen = torch.Generator().manual_seed(4)
    for nb, ids in ((4, torch.arange(n)), (r, torch.randperm(n, generator=gen))):
        layer = RelGraphConv(n, fout, r, 'basis', nb, activation=torch.relu, self_loop=self_loop, dropout=0.0)
        with torch.no_grad():
            layer.h_bias.normal_(0, 0.1)
        gout = torch.randn(n, fout, generator=gen)
        po = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
        ho = orgcn.rel_graph_conv(ids, src, dst, et, norm, po, 'basis', nb, torch.relu)
        ho.backward(gout)
        g = KGraph()
        g.add_nodes(n)
        g.add_edges(src, dst)
        layer = layer.cuda()
        hg = layer(g, ids.cuda(), et.cuda(), norm.cuda())
        hg.backward(gout.cuda())
        close(hg, ho, msg='integer-id forward')
        for k, v in layer.named_parameters():
            close(v.grad, po[k].grad, rtol=2e-4, atol_scale=2e-5, msg='integer-id grad ' + k)
    with pytest.raises(TypeError):
        RelGraphConv(16, 16, r, 'bdd', 4).cuda()(g, torch.arange(n).cuda(), et.cuda(), norm.cuda())
    with pytest.raises(ValueError):
        layer(g, (ids + 1).cuda(), et.cuda(), norm.cuda())         # an id outside [0, in_feat)


@pytest.mark.parametrize('fin,fout,act', [(200, 200, 'relu'), (200, 400, None), (64, 36, 'relu')])
def test_basis_dense_path_equals_generic_path_and_oracle(ops, fin, fout, act, monkeypatch):
    """SURVEY 8(f-3) at FB15k-237 widths: the relation-grouped MFMA path (gv_rel_rows_gemm / gv_rel_gradw_gemm + 1x1 K1
    aggregation) against the generic per-edge kernels on the same inputs, and against the CPU oracle."""
    from gcn_vae_amd.graph import KGraph
    from gcn_vae_amd.layers import RelGraphConv
    n, e, r, nb = 1500, 6000, 40, 8
    src, dst, et, norm = zipf_graph(n, e, r, seed=21)
    torch.manual_seed(3)
    activation = torch.relu if act == 'relu' else None
    layer = RelGraphConv(fin, fout, r, 'basis', nb, activation=activation, self_loop=True, dropout=0.0)
    with torch.no_grad():
        layer.h_bias.normal_(0, 0.1)
    x, gout = torch.randn(n, fin), torch.randn(n, fout)
    po = {k: v.detach().clone().requires_grad_(True) for k, v in layer.named_parameters()}
    xo = x.clone().requires_grad_(True)
    ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'basis', nb, activation)
    ho.backward(gout)
    g = KGraph()
    g.add_nodes(n)
    g.add_edges(src, dst)
    layer = layer.cuda()
    results = {}
    for mode in ('dense', 'generic'):
        monkeypatch.setenv('GV_BASIS_GENERIC', '1' if mode == 'generic' else '0')
        layer.zero_grad()
        xg = x.cuda().requires_grad_(True)
        hg = layer(g, xg, et.cuda(), norm.cuda())
        hg.backward(gout.cuda())
        results[mode] = (hg.detach(), xg.grad.detach(), {k: v.grad.detach().clone() for k, v in layer.named_parameters()})
    hd, gxd, gpd = results['dense']
    hgn, gxg, gpg = results['generic']
    close(hd, hgn, rtol=2e-5, atol_scale=2e-6, msg='dense vs generic forward')
    close(gxd, gxg, rtol=2e-5, atol_scale=2e-6, msg='dense vs generic grad_x')
    for k in gpd:
        close(gpd[k], gpg[k], rtol=1e-4, atol_scale=1e-5, msg='dense vs generic grad ' + k)
    close(hd, ho, msg='basis forward vs oracle')
    close(gxd, xo.grad, msg='basis grad_x vs oracle')
    for k, v in gpd.items():
        close(v, po[k].grad, rtol=2e-4, atol_scale=2e-5, msg='basis grad vs oracle ' + k)


def test_odd_widths_and_degenerate_graphs(ops):
    # feature widths that are not multiples of 4 take the generic kernels and the scalar GEMM loads
    n, e, r, fin, fout, nb = 70, 400, 5, 9, 15, 3
    src, dst, et, norm = zipf_graph(n, e, r, seed=1)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, fin, generator=gen)
    p = orgcn.init_params(fin, fout, r, 'bdd', nb, True, True, gen)
    gout = torch.randn(n, fout, generator=gen)
    xo = x.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ho = orgcn.rel_graph_conv(xo, src, dst, et, norm, po, 'bdd', nb, torch.relu)
    ho.backward(gout)
    gidx = ops.GraphIndex(src.cuda(), dst.cuda(), n)
    ridx = gidx.relation_index(et.cuda(), r)
    xg = x.cuda().requires_grad_(True)
    pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
    hg = ops.rel_graph_conv_bdd(xg, pg['weight'], pg['h_bias'], pg['loop_weight'], norm.cuda(), gidx, ridx, nb, 1)
    hg.backward(gout.cuda())
    close(hg, ho)
    close(xg.grad, xo.grad)
    for k in p:
        close(pg[k].grad, po[k].grad, msg=k)
    # a graph without edges: every row is act(bias + x @ loop_weight); gradients of the relation weights are zero
    empty = torch.zeros(0, dtype=torch.int64)
    g0 = ops.GraphIndex(empty.cuda(), empty.cuda(), n)
    r0 = g0.relation_index(empty.cuda(), r)
    pg = {k: v.cuda().requires_grad_(True) for k, v in p.items()}
    h0 = ops.rel_graph_conv_bdd(x.cuda(), pg['weight'], pg['h_bias'], pg['loop_weight'], None, g0, r0, nb, 1)
    close(h0, torch.relu(x @ p['loop_weight'] + p['h_bias']))
    h0.sum().backward()
    assert float(pg['weight'].grad.abs().max()) == 0.0
    # one node, self loops only
    one = torch.zeros(3, dtype=torch.int64)
    g1 = ops.GraphIndex(one.cuda(), one.cuda(), 1)
    r1 = g1.relation_index(torch.tensor([0, 1, 1]).cuda(), r)
    x1 = torch.randn(1, fin, generator=gen)
    h1 = ops.rel_graph_conv_bdd(x1.cuda(), p['weight'].cuda(), None, None, None, g1, r1, nb, 0)
    close(h1, orgcn.rel_graph_conv(x1, one, one, torch.tensor([0, 1, 1]), None, {'weight': p['weight']}, 'bdd', nb))
    # non-contiguous inputs are accepted (made contiguous), wrong dtypes are refused
    xt = torch.randn(fin, n, generator=gen).t()
    close(ops.gemm(xt.cuda(), p['loop_weight'].cuda()), xt.double() @ p['loop_weight'].double())
    with pytest.raises(TypeError):
        ops.gemm(x.cuda().double(), p['loop_weight'].cuda())


# ------------------------------------------------------------------------------------------------
# device RNG (gv_rng_fill): bit-exact against the numpy Philox restatement, fresh draws per tick
@pytest.mark.gpu
def test_device_rng_matches_philox_oracle_and_ticks(ops):
    from oracle import philox
    torch.manual_seed(1234)
    rng = ops.device_rng('cuda')
    rng.tick()                                           # also syncs the seed
    seed, tick = (int(v) for v in rng.state.cpu())
    assert seed == 1234
    n_mask, n_norm = 14541 * 200 + 3, 40 * 200 + 1      # lengths that are not multiples of 4
    keep = torch.empty(n_mask, dtype=torch.uint8, device='cuda')
    keep2 = torch.empty(1000, dtype=torch.uint8, device='cuda')
    eps = torch.empty(n_norm, dtype=torch.float32, device='cuda')
    rng.fill([(keep, ops.RNG_KEEP_MASK, 0.2, 11), (eps, ops.RNG_NORMAL, 0.0, 12), (keep2, ops.RNG_KEEP_MASK, 0.5, 13)])
    assert np.array_equal(keep.cpu().numpy(), philox.keep_mask(seed, tick, 11, n_mask, 0.2))        # bit-exact
    assert np.array_equal(keep2.cpu().numpy(), philox.keep_mask(seed, tick, 13, 1000, 0.5))
    want = philox.normals(seed, tick, 12, n_norm)
    np.testing.assert_allclose(eps.cpu().numpy(), want, rtol=0, atol=2e-5)    # libm vs device log/sincos
    assert abs(float(keep.float().mean()) - 0.8) < 2e-3
    assert abs(float(eps.mean())) < 0.05 and abs(float(eps.std()) - 1.0) < 0.05
    # drawing the same stream again within a tick forces a tick: new numbers, and the state moved by one
    first = keep.clone()
    rng.fill([(keep, ops.RNG_KEEP_MASK, 0.2, 11)])
    assert int(rng.state[1]) == tick + 1
    assert not torch.equal(first, keep)
    assert np.array_equal(keep.cpu().numpy(), philox.keep_mask(seed, tick + 1, 11, n_mask, 0.2))
    # a new torch.manual_seed value restarts the stream at tick 0
    torch.manual_seed(4321)
    rng.fill([(keep2, ops.RNG_KEEP_MASK, 0.5, 13)])
    assert [int(v) for v in rng.state.cpu()] == [4321, 0]
    assert np.array_equal(keep2.cpu().numpy(), philox.keep_mask(4321, 0, 13, 1000, 0.5))


@pytest.mark.gpu
def test_device_rng_draws_fresh_numbers_on_graph_replay(ops):
    """The embedding lookup carries the tick, so a captured forward draws new dropout masks at every replay."""
    torch.manual_seed(7)
    table = torch.randn(50, 8, device='cuda')
    ids = torch.arange(50, device='cuda')
    rng = ops.device_rng('cuda')
    keep = torch.empty(4096, dtype=torch.uint8, device='cuda')

    def step():
        ops.embedding(table, ids, rng)
        rng.fill([(keep, ops.RNG_KEEP_MASK, 0.5, 99)])

    step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append(keep.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


@pytest.mark.gpu
def test_xcd_ordered_grad_w_items_give_identical_bits(ops):
    """Reordering the grad-W work items for L2 locality (RelationIndex._xcd_order_items) changes which workgroup runs an
    item, not what it computes: same bits, including relations split over several items."""
    rs = np.random.RandomState(5)
    n, e, r, nb = 400, 12000, 16, 10
    dst = np.sort(rs.randint(0, n, e))
    src = rs.randint(0, n, e)
    et = torch.from_numpy(rs.randint(0, r, e)).cuda()
    x, g = torch.randn(n, 20, device='cuda'), torch.randn(n, 40, device='cuda')
    coef = torch.rand(e, device='cuda')
    gi = ops.GraphIndex(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n)
    outs = []
    for reorder in (False, True):
        ri = ops.RelationIndex(gi, et, r, chunk=16)
        if reorder:
            before = ri.by_rel.seg.n_items
            ri._xcd_order_items()
            assert ri.by_rel.seg.n_items >= before and int((ri.by_rel.seg.items[:, 0] >= 0).sum()) == before
        outs.append(ops.bdd_grad_weight(ri.by_rel.seg, ri.src_by_rel, ri.dst_by_rel, coef, ri.by_rel.perm, x, g, nb, 2, 4))
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------------
# evaluation scorer with the fused rank count (gv_rank_scores)
def _ranks_by_definition(ops, emb, w, a, r, b, flp):
    """#OTHER entities with a strictly larger logit + half of those that tie; scores from the same f32 MFMA GEMM."""
    q = ops.mul(emb[a].contiguous(), w[r].contiguous())
    score = ops.gemm(q, emb, trans_b=True)
    if flp is not None:
        score = score + flp
    tgt = score.gather(1, b.view(-1, 1))
    other = torch.ones_like(score, dtype=torch.bool).scatter_(1, b.view(-1, 1), False)
    return ((score > tgt) & other).sum(1).float() + 0.5 * ((score == tgt) & other).sum(1).float()


@pytest.mark.parametrize('m,v,h,flp', [(1, 5, 4, None), (37, 1000, 16, 0.25), (300, 14541, 200, -1.5), (129, 777, 200, None),
                                       (64, 64, 8, 3.0)])
def test_rank_scores_equals_materialised_count(ops, m, v, h, flp):
    gen = torch.Generator().manual_seed(m * 7 + v)
    emb = (torch.randn(v, h, generator=gen) * 0.7).cuda()
    w = torch.randn(11, h, generator=gen).cuda()
    a = torch.randint(0, v, (m,), generator=gen).cuda()
    r = torch.randint(0, 11, (m,), generator=gen).cuda()
    b = torch.randint(0, v, (m,), generator=gen).cuda()
    bias = None if flp is None else torch.tensor(flp, device='cuda')
    q = ops.mul(emb[a].contiguous(), w[r].contiguous())
    got = ops.rank_scores(q, emb, b, bias)
    want = _ranks_by_definition(ops, emb, w, a, r, b, bias)
    assert got.dtype == torch.float32 and torch.equal(got, want)
    assert float(got.min()) >= 0 and float(got.max()) < v


def test_rank_scores_ties_saturation_and_nan(ops):
    """Ties are counted, never broken in the target's favour: duplicate entity rows share the MID rank of their block; scores
    that would saturate the reference's sigmoid (all 1.0f) still rank by their logits; all-equal scores give the middle
    rank, not rank 1; a NaN target score (diverged run) ranks last; the target never counts itself."""
    h, v = 8, 130
    emb = torch.randn(v, h, generator=torch.Generator().manual_seed(0)).cuda()
    emb[7] = emb[3]                       # entity 7 duplicates entity 3
    emb[129] = emb[3]
    q = emb[[3, 3]].clone()               # query = entity 3's own row: its self score is the squared norm
    target = torch.tensor([3, 7], device='cuda')
    got = ops.rank_scores(q, emb, target)
    score = ops.gemm(q, emb, trans_b=True)
    for i in range(2):
        t = int(target[i])
        better = int((score[i] > score[i, t]).sum())
        ties = int((score[i] == score[i, t]).sum()) - 1
        assert ties == 2 and float(got[i]) == better + 0.5 * ties
    assert float(got[0]) == float(got[1])     # the three copies share one rank
    # logits of +-2500: the reference's sigmoid is exactly 1.0f / 0.0f there and everything ties; the logits still order
    big = (emb * 50.0).contiguous()
    got_big = ops.rank_scores((q * 50.0).contiguous(), big, target)
    score_big = ops.gemm((q * 50.0).contiguous(), big, trans_b=True)
    assert float(torch.sigmoid(score_big).max()) == 1.0
    assert float(got_big[0]) == float((score_big[0] > score_big[0, 3]).sum()) + 1.0
    # every candidate scores the same (e.g. collapsed embeddings): the middle rank, not the best one
    same = torch.ones(v, h, device='cuda')
    got_same = ops.rank_scores(torch.ones(4, h, device='cuda'), same, torch.tensor([0, 5, 64, 129], device='cuda'))
    assert torch.equal(got_same.cpu(), torch.full((4,), (v - 1) / 2.0))
    # NaN scores (a diverged run) never help: a NaN target ranks last, a NaN candidate counts as better than the target
    bad = emb.clone()
    bad[3] = float('nan')
    got_nan = ops.rank_scores(q, bad, target)
    sc = ops.gemm(q, bad, trans_b=True)
    assert float(got_nan[0]) == v - 1                                    # target 3's own score is NaN
    assert float(got_nan[1]) == float((sc[1] > sc[1, 7]).sum()) + 1 + 0.5  # entity 3 (NaN) better, entity 129 ties with 7
    with pytest.raises(ValueError):
        ops.rank_scores(q, emb, torch.tensor([0, v], device='cuda'))


def test_fused_ranker_matches_golden_ranks_and_unfused_path():
    from conftest import load_golden
    from gcn_vae_amd import ranking
    g = load_golden('ranking.npz')
    emb, w, flp = g['emb'].cuda(), g['w'].cuda(), g['flp'].cuda()
    trip = g['trip'].cuda()
    s, r, o = trip[:, 0], trip[:, 1], trip[:, 2]
    n = trip.shape[0]
    rs = ranking.perturb_and_get_rank(emb, w, o, r, s, n, 10, True, flp)          # vectors captured from the reference
    assert torch.equal(rs.cpu(), g['ranks_s'].to(torch.float32))       # tie-free vectors: same ranks as the reference
    for a, b in ((o, s), (s, o)):
        assert torch.equal(ranking.perturb_and_get_rank(emb, w, a, r, b, n, 10, True, flp),
                           ranking.perturb_and_get_rank_unfused(emb, w, a, r, b, n, 10, True, flp))
    first = ranking.perturb_and_get_rank(emb, w, o, r, s, n, 10, False, flp)       # quick validation: first batch only
    assert torch.equal(first, rs[:10])
    mrr = ranking.calc_mrr(emb, w, trip, hits=[1, 3, 10], eval_bz=10, flow_log_prob=flp, verbose=False)
    assert abs(mrr - float(g['mrr'])) < 1e-6
    mrr1 = ranking.calc_mrr(emb, w, trip, hits=[1], eval_bz=10, all_batches=False, flow_log_prob=flp, verbose=False)
    assert abs(mrr1 - float(g['mrr_first_batch'])) < 1e-6


def test_rank_scores_full_fb15k237_eval_properties(ops):
    """The whole FB15k-237 test split in both directions (2 x 20 466 queries x 14 541 entities x h = 200): ranks are in range,
    invariant to the launch's row chunking and to a permutation of the queries, and equal the materialised count on a sample."""
    from gcn_vae_amd import ranking
    gen = torch.Generator().manual_seed(5)
    v, h, n = 14541, 200, 20466
    emb = (torch.randn(v, h, generator=gen) * 0.3).cuda()
    w = torch.randn(237, h, generator=gen).cuda()
    trip = torch.stack([torch.randint(0, v, (n,), generator=gen), torch.randint(0, 237, (n,), generator=gen),
                        torch.randint(0, v, (n,), generator=gen)], 1).cuda()
    s, r, o = trip[:, 0], trip[:, 1], trip[:, 2]
    full = ranking.perturb_and_get_rank(emb, w, s, r, o, n, 100, True, None)
    assert full.shape == (n,) and float(full.min()) >= 0 and float(full.max()) < v
    old = ranking.MAX_QUERY_ROWS
    try:
        ranking.MAX_QUERY_ROWS = 5000
        again = ranking.perturb_and_get_rank(emb, w, s, r, o, n, 100, True, None)
    finally:
        ranking.MAX_QUERY_ROWS = old
    assert torch.equal(full, again)
    perm = torch.randperm(n, generator=gen).cuda()
    shuffled = ranking.perturb_and_get_rank(emb, w, s[perm], r[perm], o[perm], n, 100, True, None)
    assert torch.equal(shuffled, full[perm])
    pick = torch.arange(0, n, 97, device='cuda')
    assert torch.equal(full[pick], _ranks_by_definition(ops, emb, w, s[pick], r[pick], o[pick], None))


# ------------------------------------------------------------------------------------------------
def test_phase_lists_with_waves_chosen_per_item_give_the_same_rows(ops, monkeypatch):
    """PhaseOrder.build: the greedy choice of an item's wave inside its tile (GV_PHASE_GREEDY) against the snake deal -- the same tiles,
    other (wave, slot) places, shorter per-phase maxima, and an aggregation that is bit-identical (a row keeps its summation order)."""
    rs = np.random.RandomState(11)
    n, e, r = 5000, 200000, 60
    p = (np.arange(n) + 1.0) ** -1.1
    dst = np.sort(rs.choice(n, size=e, p=p / p.sum()))
    src, et = rs.randint(0, n, size=e), rs.randint(0, r, size=e)
    g = ops.GraphIndex(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n, dst_sorted=True)
    ridx = ops.RelationIndex(g, torch.from_numpy(et).cuda(), r)
    nb, si, so = 100, 5, 5
    x = torch.randn(n, nb * si, generator=torch.Generator().manual_seed(0)).cuda()
    w = torch.randn(r, nb * si * so, generator=torch.Generator().manual_seed(1)).cuda()
    outs, longest = [], []
    for greedy in (False, True):
        monkeypatch.setattr(ops.indices, 'PHASE_GREEDY', greedy)
        ph = ops.indices.PhaseOrder.build(ridx, g, 'dst', nb, si, so)
        assert ph is not None
        nw = ph.threads // 64
        lens = (ph.off[1:] - ph.off[:-1]).view(ph.n_tiles, nw, ph.n_phases)
        assert int(lens.sum()) == e
        longest.append(int(lens.max(1).values.sum()))
        outs.append(ops.bdd_aggregate_phases(ph, None, x, ops.pack_weight_phase(ph, w, nb, si, so), r, nb, si, so))
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0
    assert longest[1] < 0.95 * longest[0], longest


def test_work_items_longest_first_is_a_permutation_with_the_same_results(ops):
    """indices.largest_first: the same items in another order (long ones first, a per-batch -1-padded list untouched); an aggregation
    over the reordered list equals the one over the built list bit for bit."""
    rs = np.random.RandomState(3)
    n, e, r = 3000, 90000, 12
    p = (np.arange(n) + 1.0) ** -1.1
    dst = np.sort(rs.choice(n, size=e, p=p / p.sum()))
    src, et = rs.randint(0, n, size=e), rs.randint(0, r, size=e)
    g = ops.GraphIndex(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n, chunk=64, dst_sorted=True)
    ridx = ops.RelationIndex(g, torch.from_numpy(et).cuda(), r)
    seg = g.by_dst.seg
    lf = ops.indices.largest_first(seg)
    assert lf is not seg and ops.indices.largest_first(seg) is lf and ops.indices.largest_first(lf) is lf
    a, b = seg.items[:seg.n_items].cpu().numpy(), lf.items[:lf.n_items].cpu().numpy()
    assert sorted(map(tuple, a)) == sorted(map(tuple, b))
    size = b[:, 2] - b[:, 1]
    assert (size[:-1] >= size[1:]).all() and size[0] == 64 and size[-1] <= 1
    x = torch.randn(n, 8, generator=torch.Generator().manual_seed(0)).cuda()
    w = torch.randn(r, 4 * 2 * 2, generator=torch.Generator().manual_seed(1)).cuda()
    out0 = ops.bdd_aggregate(seg, g.nbr_by_dst, ridx.et_by_dst, None, None, x, w, 4, 2, 2)
    out1 = ops.bdd_aggregate(lf, g.nbr_by_dst, ridx.et_by_dst, None, None, x, w, 4, 2, 2)
    assert torch.equal(out0, out1) and float(out0.abs().max()) > 0
    gb = ops.GraphIndex(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), n, dst_sorted=True, sync_free=True)
    assert ops.indices.largest_first(gb.by_dst.seg) is gb.by_dst.seg          # upper-bound-sized, -1 padded: as built


# SURVEY 8(f-1) / 8(b) gv_build_csr: native, synchronisation-free index construction == the torch formulation
def _same_items(a, b):
    return (a.n_items == b.n_items and a.n_fix == b.n_fix and a.n_slots == b.n_slots and a.chunk == b.chunk
            and torch.equal(a.rowptr, b.rowptr) and torch.equal(a.items, b.items) and torch.equal(a.fix, b.fix))


@pytest.mark.parametrize('n,e,r,sorted_dst,n_src', [(50, 0, 4, True, None), (1, 7, 1, True, None), (300, 5000, 12, True, None),
                                                     (300, 5000, 12, False, None), (2000, 60000, 474, True, None),
                                                     (120, 9000, 30, True, 700), (97, 4001, 7, False, 350),
                                                     (3000, 150000, 60, False, None)])       # >= 100 000 entries: the 16-bit-key sort path
@pytest.mark.parametrize('sort', ['own', 'radix', 'merge'])
def test_native_index_build_equals_torch_formulation(ops, monkeypatch, n, e, r, sorted_dst, n_src, sort):
    # 'own': the library's kernel-only LSD radix sort (the default when ids fit 16 bits); 'radix' / 'merge': the rocPRIM
    # fallbacks for larger inputs, eager and while the stream is being captured (csrc/k_index.hip)
    monkeypatch.setenv('GV_INDEX_SORT', sort)
    rs = np.random.RandomState(n + e)
    ns = n if n_src is None else n_src
    p = (np.arange(n) + 1.0) ** -1.2
    dst = rs.choice(n, size=e, p=p / p.sum())                    # hubs: rows longer than one work item
    src = rs.randint(0, ns, size=e)
    et = rs.randint(0, r, size=e)
    if sorted_dst:
        order = np.lexsort((et, src, dst))
        src, dst, et = src[order], dst[order], et[order]
    src_t, dst_t, et_t = (torch.from_numpy(a).cuda() for a in (src, dst, et))
    built = {}
    old = ops.indices.NATIVE_INDEX
    try:
        for native in (True, False):
            ops.indices.NATIVE_INDEX = native
            g = ops.GraphIndex(src_t, dst_t, n, chunk=64, dst_sorted=sorted_dst, sync_free=True, num_src_nodes=n_src)
            built[native] = (g, ops.RelationIndex(g, et_t, r, chunk=32))
    finally:
        ops.indices.NATIVE_INDEX = old
    (gn, rn), (gt, rt) = built[True], built[False]
    assert (gn.by_dst.perm is None) == (gt.by_dst.perm is None) == sorted_dst
    if not sorted_dst:
        assert torch.equal(gn.by_dst.perm, gt.by_dst.perm)
    assert torch.equal(gn.by_src.perm, gt.by_src.perm)
    assert torch.equal(gn.nbr_by_dst, gt.nbr_by_dst) and torch.equal(gn.nbr_by_src, gt.nbr_by_src)
    assert torch.equal(gn.src32, gt.src32) and torch.equal(gn.dst32, gt.dst32)
    assert _same_items(gn.by_dst.seg, gt.by_dst.seg) and _same_items(gn.by_src.seg, gt.by_src.seg)
    for name in ('et_by_dst', 'et_by_src', 'src_by_rel', 'dst_by_rel'):
        assert torch.equal(getattr(rn, name), getattr(rt, name)), name
    assert torch.equal(rn.by_rel.perm, rt.by_rel.perm) and _same_items(rn.by_rel.seg, rt.by_rel.seg)


@pytest.mark.parametrize('sort', ['own', 'radix', 'merge'])
@pytest.mark.parametrize('T,n_ent,n_rel', [(0, 10, 3), (1, 5, 2), (5000, 300, 7), (220000, 10000, 237), (330000, 14541, 474)])
def test_native_triplet_index_equals_torch_formulation(ops, monkeypatch, T, n_ent, n_rel, sort):
    monkeypatch.setenv('GV_INDEX_SORT', sort)
    rs = np.random.RandomState(T + 1)
    p = (np.arange(n_ent) + 1.0) ** -1.0
    trip = np.stack([rs.choice(n_ent, size=T, p=p / p.sum()), rs.randint(0, n_rel, size=T), rs.randint(0, n_ent, size=T)], 1)
    trip_t = torch.from_numpy(trip.astype(np.int64)).cuda().reshape(T, 3)
    built = {}
    old = ops.indices.NATIVE_INDEX
    try:
        for native in (True, False):
            ops.indices.NATIVE_INDEX = native
            built[native] = ops.TripletIndex(trip_t, n_ent, n_rel, sync_free=True, locality=False)
    finally:
        ops.indices.NATIVE_INDEX = old
    a, b = built[True], built[False]
    for name in ('trip32', 'inc_other', 'inc_rel', 'inc_tid', 'rel_s', 'rel_o', 'rel_tid'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert _same_items(a.inc, b.inc) and _same_items(a.rel, b.rel)
    assert a.fwd_order is None and a.pos3 is None and b.fwd_order is None and b.pos3 is None


@pytest.mark.parametrize('n,e,r,sorted_dst,T', [(300, 5000, 12, True, 2000), (300, 5000, 12, False, 0), (14541, 30000, 474, True, 330000),
                                                 (2000, 60000, 474, True, 5000), (1, 7, 1, True, 3), (120, 0, 5, True, 40),
                                                 (3000, 600000, 60, False, 1000)])      # the last two: outside the batched kernels
def test_batched_index_build_equals_the_builders_one_by_one(ops, monkeypatch, n, e, r, sorted_dst, T):
    """ops.build_batch_indices (gv_triplet_lists + gv_build_csr_batch: every pass of the five orderings in one launch) against
    GraphIndex / RelationIndex / TripletIndex built one after the other: every array equal."""
    monkeypatch.delenv('GV_INDEX_SORT', raising=False)
    rs = np.random.RandomState(n + e + T)
    p = (np.arange(n) + 1.0) ** -1.2
    dst = rs.choice(n, size=e, p=p / p.sum())
    src = rs.randint(0, n, size=e)
    et = rs.randint(0, r, size=e)
    if sorted_dst:
        order = np.lexsort((et, src, dst))
        src, dst, et = src[order], dst[order], et[order]
    src_t, dst_t, et_t = (torch.from_numpy(a.astype(np.int32)).cuda() for a in (src, dst, et))
    n_trel = max(1, r // 2)
    trip = np.stack([rs.randint(0, n, size=T), rs.randint(0, n_trel, size=T), rs.randint(0, n, size=T)], 1).astype(np.int64)
    trip_t = torch.from_numpy(trip).cuda().reshape(T, 3)
    gb, rb, tb = ops.build_batch_indices(src_t, dst_t, et_t, n, r, trip_t if T else None, n, n_trel, dst_sorted=sorted_dst)
    g = ops.GraphIndex(src_t, dst_t, n, dst_sorted=sorted_dst, sync_free=True)
    ri = ops.RelationIndex(g, et_t, r)
    assert (gb.by_dst.perm is None) == sorted_dst
    if not sorted_dst:
        assert torch.equal(gb.by_dst.perm, g.by_dst.perm)
    assert torch.equal(gb.by_src.perm, g.by_src.perm)
    assert torch.equal(gb.nbr_by_dst, g.nbr_by_dst) and torch.equal(gb.nbr_by_src, g.nbr_by_src)
    assert _same_items(gb.by_dst.seg, g.by_dst.seg) and _same_items(gb.by_src.seg, g.by_src.seg)
    for name in ('et_by_dst', 'et_by_src', 'src_by_rel', 'dst_by_rel'):
        assert torch.equal(getattr(rb, name), getattr(ri, name)), name
    assert torch.equal(rb.by_rel.perm, ri.by_rel.perm) and _same_items(rb.by_rel.seg, ri.by_rel.seg)
    assert gb.relation_index(et_t, r) is rb                      # found by the layers' lookup
    if T:
        t1 = ops.TripletIndex(trip_t, n, n_trel, sync_free=True)
        for name in ('trip32', 'inc_other', 'inc_rel', 'inc_tid', 'rel_s', 'rel_o', 'rel_tid'):
            assert torch.equal(getattr(tb, name), getattr(t1, name)), name
        assert _same_items(tb.inc, t1.inc) and _same_items(tb.rel, t1.rel)
    else:
        assert tb is None


def test_gv_build_csr_batch_reports_bad_arguments(ops):
    from gcn_vae_amd import lib
    import ctypes
    jobs = (ops.batch_index._CsrJob * 9)()
    assert lib.load().gv_build_csr_batch(ctypes.addressof(jobs), 9, None, 0, None) != 0          # more orderings than a batch holds
    assert lib.load().gv_build_csr_batch(ctypes.addressof(jobs), 0, None, 0, None) == 0
    jobs[0].n, jobs[0].n_seg, jobs[0].chunk = 5, 3, 16
    assert lib.load().gv_build_csr_batch(ctypes.addressof(jobs), 1, None, 0, None) != 0 and 'NULL' in lib.last_error()
    assert lib.load().gv_triplet_lists(None, 0, 4, None, None, None, None, None, None, None, None, None) != 0
    assert lib.load().gv_widen2_i32(None, None, 3, None, None, 0, None) != 0 and lib.load().gv_widen2_i32(None, None, 0, None, None, 0, None) == 0


@pytest.mark.parametrize('n,n_seg', [(10000, 37),
                                     # the 16-bit-key path (n >= 100 000) in ONE scatter pass (n_seg <= 256) at the lengths where the
                                     # sorted-key half once ran past its area into the sort's own value input: n % 128 in [1, 64]
                                     (100000 + 33, 200), (131072 + 64, 256), (440000, 237), (100000 + 1, 3), (100000 + 57, 255),
                                     # the own sort: one tile / many tiles, 1 and 2 passes, the widest ids, a ragged last tile
                                     (1, 1), (1023, 16), (1025, 17), (70001, 513), (660000, 14541), (300007, 65536),
                                     (2200000, 40000)])
@pytest.mark.parametrize('sort', ['own', 'radix'])
def test_gv_build_csr_single_ordering(ops, monkeypatch, n, n_seg, sort):
    monkeypatch.setenv('GV_INDEX_SORT', sort)
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    rs = np.random.RandomState(3)
    chunk = 128
    keys = torch.from_numpy(rs.randint(0, n_seg, size=n).astype(np.int32)).cuda()
    ci, cf, _ = ops._index_caps(n, n_seg, chunk)
    perm, rowptr, items, fix = ops._carve_i32(keys.device, [n, n_seg + 1, 4 * ci, 4 * cf])
    ws, nbytes = ops._index_workspace(keys.device, n, n_seg)
    lib.call('gv_build_csr', ptr(keys), n, n_seg, chunk, ptr(perm), ptr(rowptr), ptr(items), ci, ptr(fix), cf, ptr(ws), nbytes,
             lib.stream())
    want_perm = torch.sort(keys.long(), stable=True)[1].to(torch.int32)
    assert torch.equal(perm, want_perm)
    counts = torch.bincount(keys.long(), minlength=n_seg)
    assert torch.equal(rowptr.long(), torch.cat([torch.zeros(1, dtype=torch.long, device='cuda'), counts.cumsum(0)]))
    it = items.view(-1, 4)
    valid = it[it[:, 0] >= 0]
    assert int((valid[:, 2] - valid[:, 1]).sum()) == n and int((valid[:, 2] - valid[:, 1]).max()) <= chunk
    assert torch.equal(valid[:, 0].long().unique(), torch.arange(n_seg, device='cuda'))       # every segment has an item
    lib_rc = lib.load().gv_build_csr(None, 5, 3, 16, None, None, None, 1, None, 1, None, 0, None)
    assert lib_rc < 0


# ------------------------------------------------------------------------------------------------
# the batch sampler's stages on the device == the host pipeline (sampling.py, pinned to the reference)
def _i32(a):
    return torch.from_numpy(np.asarray(a).astype(np.int32)).cuda().contiguous()


@pytest.mark.parametrize('n,k', [(1, 1), (7, 7), (1000, 1000), (272115, 20000), (65536, 300), (65537, 65537)])
def test_native_sampler_perm_sample_is_a_keyed_permutation(ops, n, k):
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    from oracle import philox
    out = torch.empty(k, dtype=torch.int32, device='cuda')
    lib.call('gv_perm_sample', n, k, 0x1234567890ABCDEF, 7, None, None, 0x5A01, ptr(out), lib.stream())
    got = out.cpu().numpy().astype(np.int64)
    assert len(np.unique(got)) == k and got.min() >= 0 and got.max() < n            # distinct: sampling without replacement
    if k == n:
        assert (np.sort(got) == np.arange(n)).all()
    assert np.array_equal(got, philox.perm_sample(n, k, 0x1234567890ABCDEF, 7, 0x5A01))    # the numpy restatement, bit for bit
    other = torch.empty(k, dtype=torch.int32, device='cuda')
    lib.call('gv_perm_sample', n, k, 0x1234567890ABCDEF, 8, None, None, 0x5A01, ptr(other), lib.stream())
    if n > 1000:
        assert not torch.equal(out, other)                                               # the tick selects another permutation
    # the hipGraph form: batch counter and range read from device memory when the kernel runs (tick 5 + *tick_dev 3 = 8)
    tick_dev = torch.tensor([3], dtype=torch.int64, device='cuda')
    n_dev = torch.tensor([n], dtype=torch.int32, device='cuda')
    third = torch.empty(k, dtype=torch.int32, device='cuda')
    lib.call('gv_perm_sample', 2 * n + 5, k, 0x1234567890ABCDEF, 5, ptr(tick_dev), ptr(n_dev), 0x5A01, ptr(third), lib.stream())
    assert torch.equal(third, other)


def test_native_sampler_stages_equal_host_pipeline(ops):
    """chosen triplets -> relabel (np.unique) -> negatives from GIVEN draws -> graph from the kept half: every array equals
    gcn_vae_amd.sampling's (whose outputs are pinned to vectors captured from the reference, tests/test_host_pipeline.py)."""
    from gcn_vae_amd import lib, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.lib import ptr
    data = synthetic_kg(3000, 17, 40000, seed=4)
    k, neg, n_rel, num_nodes = 5000, 6, 17, 3000
    rs = np.random.RandomState(11)
    chosen = rs.choice(len(data.train), k, replace=False)
    sub = data.train[chosen]
    st = lib.stream()
    # relabel
    uniq_v, inv = np.unique((sub[:, 0], sub[:, 2]), return_inverse=True)
    src_h, dst_h = np.reshape(inv, (2, -1))
    cap = min(2 * k, num_nodes)
    uniq, src, dst, count = (torch.empty(cap, dtype=torch.int32, device='cuda'), torch.empty(k, dtype=torch.int32, device='cuda'),
                             torch.empty(k, dtype=torch.int32, device='cuda'), torch.empty(1, dtype=torch.int32, device='cuda'))
    wb = int(lib.load().gv_relabel_workspace_bytes(num_nodes))
    ws = torch.empty(wb, dtype=torch.uint8, device='cuda')
    a_g, b_g, rel_d = _i32(sub[:, 0]), _i32(sub[:, 2]), _i32(sub[:, 1])        # held: ptr() of a temporary would dangle
    lib.call('gv_relabel_pairs', ptr(a_g), ptr(b_g), k, num_nodes, ptr(uniq), cap, ptr(src), ptr(dst),
             ptr(count), ptr(ws), wb, st)
    n = int(count.item())
    assert n == len(uniq_v) and np.array_equal(uniq[:n].cpu().numpy(), uniq_v)
    assert np.array_equal(src.cpu().numpy(), src_h) and np.array_equal(dst.cpu().numpy(), dst_h)
    # negatives: the host routine consumes numpy's global stream (randint, then uniform); hand the same draws to the device
    pos = np.stack((src_h, sub[:, 1], dst_h)).transpose()
    np.random.seed(5)
    samples_h, labels_h = sampling.negative_sampling(pos, n, neg)
    np.random.seed(5)
    values = np.random.randint(n, size=k * neg)
    coin = np.random.uniform(size=k * neg)
    samples = torch.empty(k * (neg + 1), 3, dtype=torch.int64, device='cuda')
    labels = torch.empty(k * (neg + 1), dtype=torch.float32, device='cuda')
    hit = torch.from_numpy((coin > 0.5).astype(np.uint8)).cuda()
    values_d = _i32(values)
    lib.call('gv_negative_sampling', ptr(src), ptr(rel_d), ptr(dst), k, neg, None, ptr(values_d), ptr(hit), 0, 0, None, 0,
             ptr(samples), ptr(labels), st)
    assert np.array_equal(samples.cpu().numpy(), samples_h) and np.array_equal(labels.cpu().numpy(), labels_h)
    # graph from a kept subset
    keep = rs.choice(k, size=k // 2, replace=False)
    g_h, rel_h, norm_h = sampling.build_graph_from_triplets(n, n_rel, (src_h[keep], sub[keep, 1], dst_h[keep]))
    m = len(keep)
    src2, dst2, rel2 = (torch.empty(2 * m, dtype=torch.int32, device='cuda') for _ in range(3))
    norm = torch.empty(2 * m, dtype=torch.float32, device='cuda')
    gb = int(lib.load().gv_graph_from_triplets_workspace_bytes(m, cap, n_rel))
    gws = torch.empty(gb, dtype=torch.uint8, device='cuda')
    keep_d = _i32(keep)
    lib.call('gv_graph_from_triplets', ptr(src), ptr(rel_d), ptr(dst), ptr(keep_d), m, cap, n_rel, ptr(src2),
             ptr(dst2), ptr(rel2), ptr(norm), ptr(gws), gb, st)
    es, ed = g_h.edges()
    assert np.array_equal(src2.cpu().numpy(), es.numpy()) and np.array_equal(dst2.cpu().numpy(), ed.numpy())
    assert np.array_equal(rel2.cpu().numpy(), rel_h)
    assert np.array_equal(norm.cpu().numpy(), norm_h[ed.numpy()].astype(np.float32))       # 1 / in-degree, bit for bit


@pytest.mark.parametrize('num_nodes,n_rel,k,m', [(14541, 237, 20000, 10000), (50, 3, 1, 1), (700, 9, 333, 77), (32000, 400, 30000, 30000),
                                                 (40000, 11, 9000, 4000), (5000, 5, 40000, 33000)])
def test_native_sampler_relabel_and_graph_build_over_sizes(ops, num_nodes, n_rel, k, m):
    """gv_relabel_pairs and gv_graph_from_triplets against numpy (np.unique / np.lexsort / bincount): the relabelling as one
    single-workgroup launch (ids within LDS) and as the six-launch form beyond; batch sizes of the reference's regime, odd sizes, a
    single triplet.  Integer arrays equal; the norm bit for bit."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    rs = np.random.RandomState(num_nodes + k)
    hub = (rs.zipf(1.6, size=2 * k) - 1) % num_nodes                         # a few hub entities, as in a sampled batch
    a, b = hub[:k].astype(np.int32), rs.randint(0, num_nodes, size=k).astype(np.int32)
    rel = rs.randint(0, n_rel, size=k).astype(np.int32)
    st = lib.stream()
    cap = min(2 * k, num_nodes)
    uniq, src, dst, count = (torch.empty(cap, dtype=torch.int32, device='cuda'), torch.empty(k, dtype=torch.int32, device='cuda'),
                             torch.empty(k, dtype=torch.int32, device='cuda'), torch.empty(1, dtype=torch.int32, device='cuda'))
    wb = int(lib.load().gv_relabel_workspace_bytes(num_nodes))
    ws = torch.empty(wb, dtype=torch.uint8, device='cuda')
    a_g, b_g, rel_d = _i32(a), _i32(b), _i32(rel)
    lib.call('gv_relabel_pairs', ptr(a_g), ptr(b_g), k, num_nodes, ptr(uniq), cap, ptr(src), ptr(dst), ptr(count), ptr(ws), wb, st)
    uniq_v, inv = np.unique((a, b), return_inverse=True)
    src_h, dst_h = np.reshape(inv, (2, -1))
    n = int(count.item())
    assert n == len(uniq_v) and np.array_equal(uniq[:n].cpu().numpy(), uniq_v)
    assert np.array_equal(src.cpu().numpy(), src_h) and np.array_equal(dst.cpu().numpy(), dst_h)
    keep = rs.choice(k, size=m, replace=False).astype(np.int32)
    s_k, r_k, o_k = src_h[keep], rel[keep], dst_h[keep]
    es, ed, er = np.concatenate((s_k, o_k)), np.concatenate((o_k, s_k)), np.concatenate((r_k, r_k + n_rel))
    order = np.lexsort((er, es, ed))                                         # sorted(zip(dst, src, rel)), kgvae/utils.py:146-147
    es, ed, er = es[order], ed[order], er[order]
    indeg = np.bincount(ed, minlength=cap)
    src2, dst2, rel2 = (torch.empty(2 * m, dtype=torch.int32, device='cuda') for _ in range(3))
    norm = torch.empty(2 * m, dtype=torch.float32, device='cuda')
    gb = int(lib.load().gv_graph_from_triplets_workspace_bytes(m, cap, n_rel))
    gws = torch.empty(gb, dtype=torch.uint8, device='cuda')
    keep_d = _i32(keep)
    lib.call('gv_graph_from_triplets', ptr(src), ptr(rel_d), ptr(dst), ptr(keep_d), m, cap, n_rel, ptr(src2), ptr(dst2), ptr(rel2),
             ptr(norm), ptr(gws), gb, st)
    assert np.array_equal(src2.cpu().numpy(), es) and np.array_equal(dst2.cpu().numpy(), ed) and np.array_equal(rel2.cpu().numpy(), er)
    assert np.array_equal(norm.cpu().numpy(), (1.0 / indeg[ed].astype(np.float32)).astype(np.float32))


def test_native_sampler_negative_draws_match_numpy_restatement(ops):
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    from oracle import philox
    k, neg, n_ent = 999, 4, 4321
    rs = np.random.RandomState(0)
    s, r, o = rs.randint(0, n_ent, k), rs.randint(0, 9, k), rs.randint(0, n_ent, k)
    samples = torch.empty(k * (neg + 1), 3, dtype=torch.int64, device='cuda')
    labels = torch.empty(k * (neg + 1), dtype=torch.float32, device='cuda')
    cnt = torch.tensor([n_ent], dtype=torch.int32, device='cuda')
    s_d, r_d, o_d = _i32(s), _i32(r), _i32(o)
    lib.call('gv_negative_sampling', ptr(s_d), ptr(r_d), ptr(o_d), k, neg, ptr(cnt), None, None, 99, 3, None, 0x5A02,
             ptr(samples), ptr(labels), lib.stream())
    values, hit = philox.negative_draws(k * neg, n_ent, 99, 3, 0x5A02)
    want = np.tile(np.stack([s, r, o], 1), (neg, 1))
    want[hit, 0] = values[hit]
    want[~hit, 2] = values[~hit]
    got = samples.cpu().numpy()
    assert np.array_equal(got[:k], np.stack([s, r, o], 1)) and np.array_equal(got[k:], want)
    assert np.array_equal(labels.cpu().numpy(), np.r_[np.ones(k), np.zeros(k * neg)].astype(np.float32))
    assert 0.4 < hit.mean() < 0.6 and values.min() >= 0 and values.max() < n_ent


@pytest.mark.parametrize('n,n_rel,n_trip,k', [(300, 5, 2000, 700), (2000, 11, 30000, 3000), (50, 3, 120, 120), (5000, 7, 6000, 2500)])
def test_neighborhood_sampler_equals_host_restatement_draw_for_draw(ops, n, n_rel, n_trip, k):
    """kgvae/utils.py:33-76 on the device: gv_neighborhood_sample == sampling.sample_edge_neighborhood_draws fed the same
    Philox outputs (the oracle's numpy Philox), edge for edge -- including the 'nothing seen has budget' restarts (sparse
    graphs with many components: n = 5000 / 6000 triplets) and the sample that exhausts every triplet (k = n_trip)."""
    from gcn_vae_amd import lib, sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.lib import ptr
    from oracle import philox
    data = synthetic_kg(n, n_rel, n_trip, seed=n)
    trip = data.train
    adj = sampling.adjacency_csr(n, trip)
    adj_d = [torch.from_numpy(a).cuda() for a in adj]
    nb = int(lib.load().gv_neighborhood_sample_workspace_bytes(n, len(trip)))
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    k = min(k, len(trip))
    out = torch.empty(k, dtype=torch.int32, device='cuda')
    seed, tick, stream = 0xFEDCBA9876543210, 5, 0x5A04
    lib.call('gv_neighborhood_sample', *(ptr(a) for a in adj_d), n, len(trip), k, seed, tick, None, stream, ptr(out), ptr(ws), nb,
             lib.stream())
    got = out.cpu().numpy()
    want = sampling.sample_edge_neighborhood_draws(*adj, len(trip), k, philox.neighborhood_draw(seed, tick, stream))
    assert np.array_equal(got, want)
    assert len(np.unique(got)) == k and got.min() >= 0 and got.max() < len(trip)         # distinct triplets
    # neighbourhood property: after the first pick every pick touches an already-seen vertex unless a restart happened
    seen, restarts = set(), 0
    for e in got.tolist():
        s_, o_ = int(trip[e, 0]), int(trip[e, 2])
        restarts += not (s_ in seen or o_ in seen)
        seen.update((s_, o_))
    assert restarts >= 1 and (restarts < k // 4 or n > 2 * n_trip // 3)


def test_device_sampler_neighbor_mode(ops):
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import STREAM_NBR, DeviceSampler
    from oracle import philox
    data = synthetic_kg(2000, 11, 30000, seed=1)
    sm = DeviceSampler(data.train, 2000, 11, 'cuda', seed=3, sampler='neighbor')
    b = sm.sample(3000, 0.5, 5)
    chosen = sm.last_chosen.cpu().numpy()
    want = sampling.sample_edge_neighborhood_draws(*sampling.adjacency_csr(2000, data.train), len(data.train), 3000,
                                                   philox.neighborhood_draw(sm.seed, sm.tick, STREAM_NBR))
    assert np.array_equal(chosen, want)
    ids = b.node_id.view(-1).cpu().numpy()
    samples = b.samples.cpu().numpy()
    pos_global = np.stack([ids[samples[:3000, 0]], samples[:3000, 1], ids[samples[:3000, 2]]], 1)
    assert np.array_equal(pos_global, data.train[chosen])
    with pytest.raises(ValueError):
        DeviceSampler(data.train, 2000, 11, 'cuda', sampler='random-walk')


@pytest.mark.parametrize('native', [True, False])
def test_device_sampler_batch_invariants(ops, native):
    """Either implementation: distinct sampled triplets, sorted unique node ids, (dst, src, rel)-ordered symmetric graph,
    norm = 1 / in-degree, positives present in the data, negatives differ from their positive in one endpoint."""
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import DeviceSampler
    data = synthetic_kg(2000, 11, 30000, seed=1)
    sm = DeviceSampler(data.train, 2000, 11, 'cuda', seed=3, native=native)
    b1, b2 = sm.sample(4000, 0.5, 5), sm.sample(4000, 0.5, 5)
    for b in (b1, b2):
        n = b.g.number_of_nodes()
        ids = b.node_id.view(-1).cpu().numpy()
        assert len(ids) == n and (np.diff(ids) > 0).all()
        samples, labels = b.samples.cpu().numpy(), b.labels.cpu().numpy()
        assert samples.shape == (4000 * 6, 3) and labels[:4000].all() and not labels[4000:].any()
        pos_global = np.stack([ids[samples[:4000, 0]], samples[:4000, 1], ids[samples[:4000, 2]]], 1)
        have = set(map(tuple, data.train.tolist()))
        assert all(tuple(t) in have for t in pos_global[:200].tolist())
        neg = samples[4000:].reshape(5, 4000, 3)
        same_s, same_o = neg[:, :, 0] == samples[None, :4000, 0], neg[:, :, 2] == samples[None, :4000, 2]
        assert (same_s | same_o).all() and (neg[:, :, 1] == samples[None, :4000, 1]).all() and samples.min() >= 0 and samples[:, [0, 2]].max() < n
        src, dst = (t.numpy() for t in b.g.edges())
        et, norm = b.edge_type.cpu().numpy(), b.edge_norm.view(-1).cpu().numpy()
        assert len(src) == 4000 and ((et[:, None] >= 0).all()) and et.max() < 22
        key = (dst.astype(np.int64) * n + src) * 22 + et
        assert (np.diff(key) >= 0).all()
        deg = np.bincount(dst, minlength=n)
        assert np.array_equal(norm, (1.0 / deg[dst]).astype(np.float32))
        fwd = set(zip(src[et < 11].tolist(), dst[et < 11].tolist(), et[et < 11].tolist()))
        rev = set(zip(dst[et >= 11].tolist(), src[et >= 11].tolist(), (et[et >= 11] - 11).tolist()))
        assert fwd == rev
    assert not torch.equal(b1.samples, b2.samples)


def test_segmented_graph_replays_the_recorded_program(ops):
    """segments.SegmentedGraph: kernels between two "collectives" become hipGraph segments, the collectives' closures run
    eagerly between them on every replay, a wait() directly behind its collective leaves no empty segment, tensors created
    inside a segment stay valid for later segments, and nothing captured executes during the recording pass."""
    from gcn_vae_amd import distributed as gdist
    from gcn_vae_amd.segments import SegmentedGraph
    x = torch.zeros(4096, device='cuda')
    y = torch.zeros(4096, device='cuda')
    calls = []

    def exchange():                      # stands in for an all-reduce: an eager op on the step's own tensors
        calls.append('exchange')
        y.copy_(x)
        y.mul_(2.0)

    def step():
        x.add_(1.0)                                              # segment 0
        h = gdist.start_collective(exchange)                    # eager, between segments
        t = x * 10.0                                             # segment 1 (runs beside the "collective")
        h.wait()
        z = t + y                                                # segment 2
        gdist.start_collective(lambda: calls.append('second')).wait()      # wait right behind the call: no empty segment
        return z + 0.5                                           # segment 3

    assert gdist.RECORDER is None
    sg = SegmentedGraph()
    out = sg.capture(step)
    assert gdist.RECORDER is None and calls == ['exchange', 'second']
    assert float(x.max()) == 0.0                                 # recording executed no captured kernel
    kinds = [k for k, _ in sg.actions]
    assert kinds == ['graph', 'call', 'graph', 'wait', 'graph', 'call', 'wait', 'graph'], kinds
    for i in range(1, 4):
        sg.replay()
        torch.cuda.synchronize()
        assert float(x.min()) == float(i) and float(y.min()) == 2.0 * i
        assert float(out.min()) == float(out.max()) == 10.0 * i + 2.0 * i + 0.5
    assert calls.count('exchange') == 4 and calls.count('second') == 4
    with pytest.raises(RuntimeError):                            # a failing step must not leave the recorder armed
        SegmentedGraph().capture(lambda: (_ for _ in ()).throw(RuntimeError('boom')))
    assert gdist.RECORDER is None
    x.add_(1.0)                                                  # and ordinary eager work still runs
    torch.cuda.synchronize()
    assert float(x.min()) == 4.0


@pytest.mark.gpu
@pytest.mark.parametrize('m,n,k,a_f32', [(300, 200, 200, False), (300, 200, 200, True), (257, 400, 200, False), (64, 72, 64, True),
                                         (130, 96, 456, False), (1000, 100, 224, True), (33, 64, 8, False), (90, 68, 40, False),
                                         (70, 36, 232, True), (500, 132, 200, False), (230, 200, 1000, False), (400, 232, 776, False)])
def test_gemm_bf16_nt_every_epilogue_and_ragged_column_counts(ops, m, n, k, a_f32):
    """gv_gemm_bf16_nt against fp32 matmul of the bf16-rounded operands (the semantics oracle/bf16.py states): ragged row /
    column / depth counts around the 64-wide tiles, each epilogue output (fp32, fp32 accumulate, bf16, transposed bf16, bias,
    ReLU, ReLU mask) and split-K (the two deep cases take the whole-output kernel of the weight-gradient products)."""
    from oracle import bf16 as obf
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m * 1000 + n)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(n, k, generator=g) * 0.2
    bias = torch.randn(n, generator=g)
    maskv = torch.randn(m, n, generator=g)
    ref = obf._r(a) @ obf._r(b).t()
    a_d = a.to(dev) if a_f32 else a.to(dev).to(torch.bfloat16)
    b_d = b.to(dev).to(torch.bfloat16)
    ldt = (m + 3) // 4 * 4

    def outs():
        return (torch.full((m, n), 7.0, device=dev), torch.zeros(m, n, dtype=torch.bfloat16, device=dev),
                torch.zeros(n, ldt, dtype=torch.bfloat16, device=dev))
    # plain: fp32 + bf16 + transposed bf16
    c, cb, ct = outs()
    ops.gemm_bf16_nt(a_d, b_d, m, n, k, c_f32=c, c_bf16=cb, c_bf16_t=ct)
    close(c, ref, rtol=1e-4, atol_scale=1e-5, msg='plain')
    assert torch.equal(cb.float(), c.to(torch.bfloat16).float()) and torch.equal(ct[:, :m].float().t(), cb.float())
    # bias + relu, accumulate
    c, cb, ct = outs()
    ops.gemm_bf16_nt(a_d, b_d, m, n, k, bias=bias.to(dev), relu=True, c_f32=c, accumulate=True)
    close(c, torch.relu(ref + bias) + 7.0, rtol=1e-4, atol_scale=1e-5, msg='bias relu accumulate')
    # ReLU mask (kept where mask > 0), every output
    c, cb, ct = outs()
    mk = maskv.to(dev).to(torch.bfloat16)
    ops.gemm_bf16_nt(a_d, b_d, m, n, k, mask=mk, c_f32=c, c_bf16=cb, c_bf16_t=ct)
    want = torch.where(mk.float().cpu() > 0, ref, torch.zeros(()))
    close(c, want, rtol=1e-4, atol_scale=1e-5, msg='masked')
    assert torch.equal(cb.float(), c.to(torch.bfloat16).float()) and torch.equal(ct[:, :m].float().t(), cb.float())
    # split-K (fp32 result only), with accumulate
    for split in (2, 3):
        c = torch.full((m, n), -3.0, device=dev)
        ops.gemm_bf16_nt(a_d, b_d, m, n, k, c_f32=c, accumulate=True, split_k=split)
        close(c, ref - 3.0, rtol=1e-4, atol_scale=1e-5, msg=f'split-k {split}')


@pytest.mark.gpu
@pytest.mark.parametrize('m,widths,masked', [(300, [200, 200, 200, 400], False), (14541, [200, 200, 200, 200, 200, 400], False),
                                             (257, [400, 200, 200, 200], True), (64, [72, 40, 136, 8], False),
                                             (1, [8, 8], True), (130, [400, 264, 24, 200], True), (1000, [200, 400], False)])
def test_made_chain_is_bit_identical_to_the_product_by_product_launches(ops, m, widths, masked):
    """gv_made_chain (one launch for a chain of NT products, activations in LDS, fragment-packed weights) against the same
    chain on gv_gemm_bf16_nt launches: every stored tensor bit for bit -- forward form (bias, ReLU, bf16 + transposed copies,
    fp32 head) and backward form (ReLU masks of stored activations, accumulating fp32 tail) -- and against the fp32 matmul of
    the bf16-rounded operands."""
    from oracle import bf16 as obf
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + len(widths))
    L = len(widths) - 1
    x = torch.randn(m, widths[0], generator=g).to(dev).to(torch.bfloat16)
    ws = [(torch.randn(widths[i + 1], widths[i], generator=g) * (1.5 / widths[i] ** 0.5)).to(dev) for i in range(L)]
    bs = [None if masked else torch.randn(widths[i + 1], generator=g).to(dev) for i in range(L)]
    masks = [torch.randn(m, widths[i + 1], generator=g).to(dev).to(torch.bfloat16) if masked else None for i in range(L)]
    mp = (m + 7) // 8 * 8

    def buffers():
        ob = [torch.zeros(m, (widths[i + 1] + 7) // 8 * 8, dtype=torch.bfloat16, device=dev) for i in range(L - 1)]
        ot = [torch.zeros(widths[i + 1], mp, dtype=torch.bfloat16, device=dev) for i in range(L - 1)]
        of = torch.full((m, widths[L]), 0.25, device=dev)
        return ob, ot, of
    # product by product
    ob1, ot1, of1 = buffers()
    inp = x
    for i in range(L):
        wb = ws[i].to(torch.bfloat16)
        last = i == L - 1
        ops.gemm_bf16_nt(inp, wb, m, widths[i + 1], widths[i], bias=bs[i], relu=not masked and not last, mask=masks[i] if not last else None,
                         c_f32=of1 if last else None, accumulate=last and masked, c_bf16=None if last else ob1[i],
                         c_bf16_t=None if last else ot1[i])
        inp = None if last else ob1[i]
    # one launch
    assert ops.made_chain_fits(widths[1:], widths[:-1], masked)
    ob2, ot2, of2 = buffers()
    layers = []
    for i in range(L):
        last = i == L - 1
        pf, pb = ops.made_pack_weight(ws[i], bwd=False)
        assert pb is None
        layers.append(dict(w_packed=pf, n=widths[i + 1], k=widths[i], bias=bs[i], relu=not masked and not last,
                           mask=masks[i] if not last else None, out_f32=of2 if last else None, accumulate=last and masked,
                           out_bf16=None if last else ob2[i], out_bf16_t=None if last else ot2[i]))
    ops.made_chain(x, m, layers)
    torch.cuda.synchronize()
    assert torch.equal(of1, of2)
    for a, b in zip(ob1 + ot1, ob2 + ot2):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    if masked:
        # the form the fused MADE backward uses: masks given TRANSPOSED (a forward chain's out_bf16_t), no row-major copies,
        # and the last layer adding add_src in the columns whose count is 0 instead of accumulating
        ob3, ot3, of3 = buffers()
        src = torch.randn(m, widths[L], generator=g).to(dev)
        cnt = torch.randint(0, 2, (widths[L],), generator=g).to(torch.int32).to(dev)
        layers_t = []
        for i in range(L):
            last = i == L - 1
            mt = None
            if not last:
                mt = torch.zeros(widths[i + 1], mp, dtype=torch.bfloat16, device=dev)
                mt[:, :m] = masks[i].t()
            layers_t.append(dict(w_packed=layers[i]['w_packed'], n=widths[i + 1], k=widths[i], mask_t=mt,
                                 out_f32=of3 if last else None, add=(src, cnt) if last else None,
                                 out_bf16_t=None if last else ot3[i]))
        ops.made_chain(x, m, layers_t)
        torch.cuda.synchronize()
        plain = torch.empty(m, widths[L], device=dev)
        ops.made_chain(x, m, [dict(l_, out_bf16=None, out_bf16_t=None) for l_ in layers[:-1]] + [dict(layers[-1], out_f32=plain, accumulate=False)])
        want = plain + torch.where(cnt == 0, src, torch.zeros((), device=dev))
        assert torch.equal(of3, want)
        for a, b in zip(ot2, ot3):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        # ... and with the masks as sign BITS (gv_chain_layer.mask_bits: word [row][tile of 32 columns])
        def words(t):
            nt = (t.shape[1] + 31) // 32
            pos = torch.zeros(t.shape[0], nt * 32, dtype=torch.int64, device=dev)
            pos[:, :t.shape[1]] = (t.float() > 0).long()
            w = (pos.view(t.shape[0], nt, 32) << torch.arange(32, device=dev)).sum(dim=2)
            return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
        ob4, ot4, of4 = buffers()
        layers_b = [dict(l_, mask_t=None, mask_bits=None if i == L - 1 else words(masks[i]), out_bf16_t=None if i == L - 1 else ot4[i],
                         out_f32=of4 if i == L - 1 else None) for i, l_ in enumerate(layers_t)]
        ops.made_chain(x, m, layers_b)
        torch.cuda.synchronize()
        assert torch.equal(of4, want)
        for a, b in zip(ot2, ot4):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    else:
        # sign bits written by a forward chain (out_bits) = the signs of its stored bf16 activations
        bits = [torch.zeros(m, (widths[i + 1] + 31) // 32 + 1, dtype=torch.int32, device=dev) for i in range(L - 1)]
        ob5, ot5, of5 = buffers()
        ops.made_chain(x, m, [dict(l_, out_bf16=None, out_bits=bits[i] if i < L - 1 else None, out_bf16_t=ot5[i] if i < L - 1 else None,
                                   out_f32=of5 if i == L - 1 else None) for i, l_ in enumerate(layers)])
        torch.cuda.synchronize()
        assert torch.equal(of5, of2)
        for i in range(L - 1):
            n_i = widths[i + 1]
            got = ((bits[i][:, :(n_i + 31) // 32].long().unsqueeze(2) >> torch.arange(32, device=dev)) & 1).reshape(m, -1)[:, :n_i]
            assert torch.equal(got.bool(), ob2[i][:, :n_i].float() > 0)
            assert bool((bits[i][:, (n_i + 31) // 32:] == 0).all())            # the pad word is not touched
            if n_i % 32:
                assert bool((got.new_tensor(0) == ((bits[i][:, (n_i - 1) // 32].long() & 0xffffffff) >> (n_i % 32))).all())
    # and the arithmetic itself
    ref = x.float().cpu()
    for i in range(L):
        ref = ref @ obf._r(ws[i].cpu()).t()
        if bs[i] is not None:
            ref = ref + bs[i].cpu()
        if i < L - 1:
            ref = torch.relu(ref) if not masked else torch.where(masks[i].float().cpu() > 0, ref, torch.zeros(()))
            ref = obf._r(ref)
    close(of2, ref + (0.25 if masked else 0.0), rtol=2e-3, atol_scale=2e-3, msg='chain vs fp32 matmul of rounded operands')


@pytest.mark.gpu
@pytest.mark.parametrize('m,d,hidden,n_hidden,with_copies', [(300, 200, 200, 2, True), (14541, 200, 200, 3, True), (65, 8, 8, 1, True),
                                                            (1000, 200, 264, 1, False), (130, 24, 40, 2, True)])
def test_made_chain_with_the_iaf_update_in_its_last_layer_is_bit_identical_to_the_separate_update(ops, m, d, hidden, n_hidden,
                                                                                                 with_copies):
    """The IAF update fused into the chain's last layer (kgvae/flow_network.py:92-96; gv_chain_layer.iaf_*) against the same chain
    storing [mu | alpha] followed by gv_iaf_update_fwd_bf16: x_new (fp32, bf16 row-major, bf16 transposed) bit for bit, with
    columns of count 0 passed through; exp(alpha + mu) and alpha equal what the stored [mu | alpha] gives."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + d)
    widths = [d] + [hidden] * (n_hidden + 1) + [2 * d]
    L = len(widths) - 1
    x_old = torch.randn(m, d, generator=g).to(dev)
    xb = x_old.to(torch.bfloat16)
    z = torch.randn(m, d, generator=g).to(dev)
    ws = [(torch.randn(widths[i + 1], widths[i], generator=g) * (1.0 / widths[i] ** 0.5)).to(dev) for i in range(L)]
    bs = [torch.randn(widths[i + 1], generator=g).to(dev) * 0.1 for i in range(L)]
    cc = torch.randint(0, 3, (d,), generator=g).to(torch.int32).to(dev)
    cc[-1] = 0                                              # the reference's middle passes leave the last column alone
    mp = (m + 7) // 8 * 8
    dp = (d + 7) // 8 * 8
    packed = ops.made_pack_weights(ws)
    hidden_layers = [dict(w_packed=packed[i][0], n=widths[i + 1], k=widths[i], bias=bs[i], relu=True) for i in range(L - 1)]
    # separate: chain -> [mu | alpha] -> update kernel
    net = torch.empty(m, 2 * d, device=dev)
    ops.made_chain(xb, m, hidden_layers + [dict(w_packed=packed[L - 1][0], n=2 * d, k=widths[L - 1], bias=bs[L - 1], out_f32=net)])
    x1, x1b, x1t = torch.zeros(m, d, device=dev), torch.zeros(m, dp, dtype=torch.bfloat16, device=dev), \
        torch.zeros(d, mp, dtype=torch.bfloat16, device=dev)
    lib.call('gv_iaf_update_fwd_bf16', ptr(z), ptr(net), 2 * d, ptr(x_old), ptr(cc), ptr(x1), ptr(x1b), x1b.stride(0), ptr(x1t),
             x1t.stride(0), m, d, lib.stream())
    # fused
    packed_iaf = ops.made_pack_weights(ws, iaf_last=True)
    for i in range(L - 1):
        assert torch.equal(packed[i][0].view(torch.int16), packed_iaf[i][0].view(torch.int16))
    for i in range(L):
        assert torch.equal(packed[i][1].view(torch.int16), packed_iaf[i][1].view(torch.int16))     # the transposed copies stay plain
    assert torch.equal(ops.made_pack_weight_iaf(ws[L - 1]).view(torch.int16), packed_iaf[L - 1][0].view(torch.int16))
    x2, x2b, x2t = torch.zeros(m, d, device=dev), torch.zeros(m, dp, dtype=torch.bfloat16, device=dev), \
        torch.zeros(d, mp, dtype=torch.bfloat16, device=dev)
    ex, alpha, net2 = torch.zeros(m, d, device=dev), torch.zeros(m, d, device=dev), torch.zeros(m, 2 * d, device=dev)
    last = dict(w_packed=packed_iaf[L - 1][0], n=2 * d, k=widths[L - 1], bias=bs[L - 1], out_f32=net2,
                iaf=dict(z=z, x_old=x_old, colcount=cc, x_new=x2, ex=ex, alpha=alpha))
    if with_copies:
        last.update(out_bf16=x2b, out_bf16_t=x2t)
    ops.made_chain(xb, m, hidden_layers + [last])
    torch.cuda.synchronize()
    assert torch.equal(net, net2)
    assert torch.equal(x1, x2)
    assert bool((x2[:, cc == 0] == x_old[:, cc == 0]).all())
    if with_copies:
        assert torch.equal(x1b.view(torch.int16), x2b.view(torch.int16)) and torch.equal(x1t.view(torch.int16), x2t.view(torch.int16))
    assert torch.equal(alpha, net[:, d:])
    # exp(alpha + mu) is what the backward kernel recomputes from [mu | alpha]: x_new = z * ex where the column is updated
    upd = cc > 0
    assert torch.equal((z * ex)[:, upd], x1[:, upd])
    # x_new alone (the last pass: no operand copies, no [mu | alpha])
    x3 = torch.zeros(m, d, device=dev)
    ops.made_chain(xb, m, hidden_layers + [dict(w_packed=packed_iaf[L - 1][0], n=2 * d, k=widths[L - 1], bias=bs[L - 1],
                                                iaf=dict(z=z, x_old=x_old, colcount=cc, x_new=x3))])
    assert torch.equal(x3, x1)
    # fp32 x_new only where the NEXT pass hands a column through (groups of 4 columns holding a count of 0)
    nxt = torch.ones(d, dtype=torch.int32, device=dev)
    nxt[-1] = 0
    if d > 8:
        nxt[5] = 0
    x4 = torch.full((m, d), -7.0, device=dev)
    ops.made_chain(xb, m, hidden_layers + [dict(w_packed=packed_iaf[L - 1][0], n=2 * d, k=widths[L - 1], bias=bs[L - 1],
                                                iaf=dict(z=z, x_old=x_old, colcount=cc, x_new=x4, keep=nxt))])
    kept = (nxt.view(-1, 4) == 0).any(dim=1).repeat_interleave(4)
    assert torch.equal(x4[:, kept], x1[:, kept]) and bool((x4[:, ~kept] == -7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize('n,d,with_ld', [(1, 8, True), (300, 40, False), (14541, 200, True), (4099, 200, False), (257, 1024, True)])
def test_iaf_update_backward_of_the_broadcast_row_pass(ops, n, d, with_ld):
    """gv_iaf_update_bwd_row0 (pass 0 of a MADE backward: one [mu | alpha] row for every node; kgvae/flow_network.py:85-98)
    against the generic update backward + column sums + axpby it replaces: g_z accumulated, the row's gradient."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n + d)
    z, gx, gz0 = (torch.randn(n, d, generator=g).to(dev) for _ in range(3))
    row = (torch.randn(1, 2 * d, generator=g) * 0.5).to(dev)
    cc = torch.randint(0, 3, (d,), generator=g).to(torch.int32).to(dev)
    gld = torch.randn(n, generator=g).to(dev) if with_ld else None
    old = ops.MADE_ROW0_BWD
    res = {}
    try:
        for on in (True, False):
            ops.MADE_ROW0_BWD = on
            gz = gz0.clone()
            res[on] = (ops.iaf_bwd_row0(z, row, cc, gx, gld, gz), gz)
    finally:
        ops.MADE_ROW0_BWD = old
    torch.cuda.synchronize()
    assert torch.equal(res[True][1], res[False][1])                 # per element the same expressions
    close(res[True][0], res[False][0], rtol=2e-5, atol_scale=2e-6, msg='row gradient (column sums in another fixed order)')
    # and against the definition
    e = torch.exp(row[0, d:] + row[0, :d]).double()
    gc = gx.double() * cc.double()
    g_mu = torch.where(cc > 0, gc * z.double() * e, torch.zeros((), dtype=torch.float64, device=dev))
    want = torch.cat([g_mu.sum(0), g_mu.sum(0) + (gld.double().sum() if with_ld else 0.0)])
    close(res[True][0].view(-1), want.float(), rtol=2e-4, atol_scale=2e-5, msg='row gradient vs float64')


@pytest.mark.gpu
@pytest.mark.parametrize('m,d', [(300, 200), (65, 8), (4099, 24)])
def test_made_chain_half_width_input_and_the_update_backward_that_writes_it(ops, m, d):
    """gv_chain_layer.x_dup_half (the chain stages the g_mu half of [g_mu | g_alpha] twice) against the full-width input, and
    gv_iaf_update_bwd_bf16_ex's flags: bit 1 writes the g_mu half alone (valid without a log-det gradient: g_alpha == g_mu),
    bit 0 starts g_z instead of adding to it."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + d)
    z, ex, gx, gz0 = (torch.randn(n_, d, generator=g).to(dev) for n_ in (m, m, m, m))
    ex = ex.exp()
    cc = torch.randint(0, 3, (d,), generator=g).to(torch.int32).to(dev)
    mp = (m + 7) // 8 * 8
    bf = dict(dtype=torch.bfloat16, device=dev)
    out = {}
    for flags in (0, 1, 2, 3):
        gz = gz0.clone()
        gb, gt = torch.zeros(m, 2 * d, **bf), torch.zeros(2 * d, mp, **bf)
        lib.call('gv_iaf_update_bwd_bf16_ex', ptr(z), ptr(ex), d, ptr(cc), ptr(gx), None, ptr(gz), ptr(gb), gb.stride(0), ptr(gt),
                 gt.stride(0), None, flags, m, d, lib.stream())
        out[flags] = (gz, gb, gt)
    torch.cuda.synchronize()
    assert torch.equal(out[0][1][:, :d].view(torch.int16), out[0][1][:, d:].view(torch.int16)) or bool((out[0][1][:, :d].float() == out[0][1][:, d:].float()).all())
    for flags in (1, 2, 3):
        assert torch.equal(out[flags][2].view(torch.int16), out[0][2].view(torch.int16))                 # the transposed copy is always whole
        assert torch.equal(out[flags][1][:, :d].view(torch.int16), out[0][1][:, :d].view(torch.int16))
    assert bool((out[2][1][:, d:] == 0).all()) and bool((out[3][1][:, d:] == 0).all())                 # alpha half untouched
    assert torch.equal(out[0][0], out[2][0]) and torch.equal(out[1][0], out[3][0])
    assert torch.equal(out[0][0], gz0 + out[1][0])                                                       # written, not added
    # the chain on the half-width input == the chain on [g_mu | g_mu]
    widths = [2 * d, d, d]
    ws = [(torch.randn(widths[i + 1], widths[i], generator=g) / widths[i] ** 0.5).to(dev) for i in range(2)]
    packed = [ops.made_pack_weight(w, bwd=False)[0] for w in ws]
    full = torch.cat([out[2][1][:, :d], out[2][1][:, :d]], dim=1).contiguous()
    res = []
    for x, dup in ((full, False), (out[2][1], True)):
        o = torch.empty(m, d, device=dev)
        ops.made_chain(x, m, [dict(w_packed=packed[0], n=d, k=2 * d, relu=True, x_dup_half=dup), dict(w_packed=packed[1], n=d, k=d, out_f32=o)])
        res.append(o)
    assert torch.equal(res[0], res[1])


def _untile(t, rows):
    """[tiles][cols][64] -> [cols][rows]: the plain transposed form of a copy stored in tiles of 64 rows."""
    return t.permute(1, 0, 2).reshape(t.shape[1], -1)[:, :rows]


@pytest.mark.gpu
@pytest.mark.parametrize('m,d', [(300, 200), (64, 24), (1000, 72)])
def test_transposed_copies_in_tiles_of_64_rows_hold_the_same_values_and_the_product_reads_them(ops, m, d):
    """The 64-row-tile form of the transposed bf16 copies (gv_chain_layer.t_tile, gv_iaf_update_fwd_bf16_tiles,
    gv_iaf_update_bwd_bf16_ex flag 4) against the [column][row] form of the same launches, bit for bit; the rows behind m of the
    last tile are written as ZEROS (they take part in the weight-gradient reduction: the caller's buffer needs no fill), columns
    outside the destination stay untouched;
    gv_gemm_bf16_gradw_tiles on tiled operands against gv_gemm_bf16_gradw on the same operands as [row][k], bit for bit."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m * 3 + d)
    bf = dict(dtype=torch.bfloat16, device=dev)
    T, mp = (m + 63) // 64, (m + 7) // 8 * 8
    extra = 16                                                # the buffers are column ranges of a wider allocation
    i16 = lambda t: t.contiguous().view(torch.int16)
    # forward update
    z, xold = (torch.randn(m, d, generator=g).to(dev) for _ in range(2))
    net = (torch.randn(m, 2 * d, generator=g) * 0.3).to(dev)
    cnt = torch.randint(0, 3, (d,), generator=g).to(torch.int32).to(dev)
    xn, xb, xt = torch.empty(m, d, device=dev), torch.zeros(m, d, **bf), torch.zeros(d, mp, **bf)
    lib.call('gv_iaf_update_fwd_bf16', ptr(z), ptr(net), 2 * d, ptr(xold), ptr(cnt), ptr(xn), ptr(xb), d, ptr(xt), mp, m, d, lib.stream())
    xn2, xb2 = torch.empty(m, d, device=dev), torch.zeros(m, d, **bf)
    big = torch.full((T, d + extra, 64), 7.0, **bf)
    lib.call('gv_iaf_update_fwd_bf16_tiles', ptr(z), ptr(net), 2 * d, ptr(xold), ptr(cnt), ptr(xn2), ptr(xb2), d, ptr(big[:, extra:]),
             (d + extra) * 64, m, d, lib.stream())
    assert torch.equal(xn, xn2) and torch.equal(i16(xb), i16(xb2))
    assert torch.equal(i16(_untile(big[:, extra:], m)), i16(xt[:, :m]))
    assert bool((big[:, :extra] == 7.0).all()) and (m % 64 == 0 or bool((big[T - 1, extra:, m % 64:] == 0.0).all()))
    # backward update
    ex, gx = torch.randn(m, d, generator=g).to(dev).exp(), torch.randn(m, d, generator=g).to(dev)
    gld = torch.randn(m, generator=g).to(dev)
    res = []
    for tiles in (False, True):
        gz, gb = torch.ones(m, d, device=dev), torch.zeros(m, 2 * d, **bf)
        gt = torch.full((T, 2 * d + extra, 64), 7.0, **bf) if tiles else torch.zeros(2 * d, mp, **bf)
        lib.call('gv_iaf_update_bwd_bf16_ex', ptr(z), ptr(ex), d, ptr(cnt), ptr(gx), ptr(gld), ptr(gz), ptr(gb), 2 * d,
                 ptr(gt[:, extra:] if tiles else gt), (2 * d + extra) * 64 if tiles else mp, None, 4 if tiles else 0, m, d, lib.stream())
        res.append((gz, gb, gt))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(i16(res[0][1]), i16(res[1][1]))
    assert torch.equal(i16(_untile(res[1][2][:, extra:], m)), i16(res[0][2][:, :m]))
    assert bool((res[1][2][:, :extra] == 7.0).all()) and (m % 64 == 0 or bool((res[1][2][T - 1, extra:, m % 64:] == 0.0).all()))
    # a chain: hidden layer and a last layer that carries the update, both with a transposed copy
    k0 = 40
    ws = [(torch.randn(d, k0, generator=g) / k0 ** 0.5).to(dev), (torch.randn(2 * d, d, generator=g) / d ** 0.5).to(dev)]
    bs = [torch.randn(d, generator=g).to(dev) * 0.1, torch.randn(2 * d, generator=g).to(dev) * 0.1]
    x = torch.randn(m, k0, generator=g).to(dev).to(torch.bfloat16)
    packed = ops.made_pack_weights(ws, iaf_last=True)
    outs = []
    for tiles in (False, True):
        if tiles:
            buf = torch.full((T, 2 * d + extra, 64), 7.0, **bf)
            t0, t1 = dict(out_bf16_t=buf[:, extra:extra + d], t_tile=(2 * d + extra) * 64), dict(out_bf16_t=buf[:, extra + d:], t_tile=(2 * d + extra) * 64)
        else:
            buf = torch.zeros(2 * d, mp, **bf)
            t0, t1 = dict(out_bf16_t=buf[:d]), dict(out_bf16_t=buf[d:])
        x1, x1b = torch.empty(m, d, device=dev), torch.zeros(m, d, **bf)
        ops.made_chain(x, m, [dict(w_packed=packed[0][0], n=d, k=k0, bias=bs[0], relu=True, **t0),
                              dict(w_packed=packed[1][0], n=2 * d, k=d, bias=bs[1], iaf=dict(z=z, x_old=xold, colcount=cnt, x_new=x1),
                                   out_bf16=x1b, **t1)])
        outs.append((x1, x1b, buf))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(i16(outs[0][1]), i16(outs[1][1]))
    assert torch.equal(i16(_untile(outs[1][2][:, extra:], m)), i16(outs[0][2][:, :m]))
    assert bool((outs[1][2][:, :extra] == 7.0).all()) and (m % 64 == 0 or bool((outs[1][2][T - 1, extra:, m % 64:] == 0.0).all()))
    # the weight-gradient product on the tiled copies of two passes == on the same operands as [row][k]
    S = 2
    tl = torch.zeros(S * T, 3 * d + extra, 64, **bf)
    tl[:, :, :] = torch.randn(S * T, 3 * d + extra, 64, generator=g).to(dev).to(torch.bfloat16)
    if m % 64:
        tl.view(S, T, -1, 64)[:, T - 1, :, m % 64:] = 0
    a_t, b_t = tl[:, extra:extra + 2 * d], tl[:, extra + 2 * d:]
    k = S * T * 64
    split = 2 if k < 512 else 3
    if ops.gemm_bf16_gradw_fits(2 * d, d, k, split):
        a_p, b_p = _untile(a_t, k).contiguous(), _untile(b_t, k).contiguous()
        c0, r0 = torch.full((2 * d, d), 0.5, device=dev), torch.full((2 * d,), -1.0, device=dev)
        c1, r1 = c0.clone(), r0.clone()
        ops.gemm_bf16_gradw(a_p, b_p, 2 * d, d, k, c0, accumulate=True, a_rowsum=r0, split_k=split)
        ops.gemm_bf16_gradw_tiles(a_t, (3 * d + extra) * 64, b_t, (3 * d + extra) * 64, 2 * d, d, k, c1, accumulate=True, a_rowsum=r1, split_k=split)
        assert torch.equal(c0, c1) and torch.equal(r0, r1)
        close(c1, a_p.float().cpu() @ b_p.float().cpu().t() + 0.5, rtol=1e-4, atol_scale=1e-5, msg='c')
    else:
        assert k < 128 * split


@pytest.mark.gpu
@pytest.mark.parametrize('cap,live,n,k', [(1000, 700, 200, 200), (14541, 10211, 400, 200), (300, 300, 72, 40), (257, 0, 64, 64)])
def test_gemm_skips_the_padding_rows_of_a_static_shape_batch(ops, cap, live, n, k):
    """gv_gemm_f32_live_rows through ops.live_rows: rows of a row-major A behind *rows_dev are padding -- whole 64-row tiles of them
    come out as zeros without being computed, every row before them as from the plain product (bit for bit); for A stored [K, M]
    (the weight-gradient form) the reduction stops at *rows_dev."""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(cap + live)
    a = torch.randn(cap, k, generator=g).to(dev)
    b = torch.randn(k, n, generator=g).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    rows = torch.tensor([live], dtype=torch.int32, device=dev)
    plain = ops.gemm(a, b, bias=bias, act=ops.ACT_RELU)
    with ops.live_rows(rows, cap):
        got = ops.gemm(a, b, bias=bias, act=ops.ACT_RELU)
        other = ops.gemm(a[:cap - 1], b, bias=bias)                    # another row count: not a node array, untouched
    first_skipped = (live + 63) // 64 * 64
    assert torch.equal(got[:first_skipped], plain[:first_skipped])
    assert bool((got[first_skipped:] == 0).all())
    assert torch.equal(other, ops.gemm(a[:cap - 1], b, bias=bias))
    # weight-gradient form: x^T g over the rows that exist
    gr = torch.randn(cap, n, generator=g).to(dev)
    for split in (1, ops.pick_split_k(k, n, cap)):
        want = ops.gemm(a[:live].contiguous(), gr[:live].contiguous(), trans_a=True, split_k=max(1, split)) if live else torch.zeros(k, n, device=dev)
        with ops.live_rows(rows, cap):
            got_w = ops.gemm(a, gr, trans_a=True, split_k=max(1, split))
        close(got_w, want, rtol=1e-5, atol_scale=1e-6, msg=f'x^T g, split_k={split}')
    assert ops.LIVE_ROWS is None


@pytest.mark.gpu
def test_made_pack_weight_transposed_form_equals_packing_the_transpose(ops):
    dev = torch.device('cuda:0')
    w = torch.randn(136, 72, device=dev)
    pf, pb = ops.made_pack_weight(w)
    pt, _ = ops.made_pack_weight(w.t().contiguous(), bwd=False)
    assert torch.equal(pb.view(torch.int16), pt.view(torch.int16))
    # layout: [tile of 32 rows][16-deep step][lane][8]
    ks = (72 + 15) // 16
    frag = pf.view(-1, ks, 64, 8).float().cpu()
    wb = w.to(torch.bfloat16).float().cpu()
    for t, s, lane in ((0, 0, 0), (1, 2, 37), (4, 4, 63), (4, 4, 7)):
        row, c0 = 32 * t + (lane & 31), 16 * s + 8 * (lane >> 5)
        want = torch.zeros(8)
        if row < 136:
            seg = wb[row, c0:min(c0 + 8, 72)]
            want[:seg.numel()] = seg
        assert torch.equal(frag[t, s, lane], want), (t, s, lane)


@pytest.mark.gpu
@pytest.mark.parametrize('widths', [[200, 200, 200, 200, 200, 400], [16, 8, 32], [500, 512, 24, 400], [260, 300, 4, 256]])
def test_made_row_chain_matches_bf16_operand_products(ops, widths):
    """gv_made_row_fwd / gv_made_row_bwd (MADE's pass 0: the masked MLP on ONE row, one workgroup) against the same chain
    written with fp32 matmuls of bf16-rounded operands (oracle/bf16.py's semantics): activations, the row gradients, the
    outer-product weight gradients and the bias gradients."""
    from oracle import bf16 as obf
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(sum(widths))
    L = len(widths) - 1
    ws = [(torch.randn(widths[i + 1], widths[i], generator=g) * (1.5 / widths[i] ** 0.5)) for i in range(L)]
    bs = [torch.randn(widths[i + 1], generator=g) * 0.3 for i in range(L)]
    x = torch.randn(1, widths[0], generator=g)
    gy = torch.randn(1, widths[L], generator=g)
    # reference
    with obf.enabled():
        xr = x.clone().requires_grad_(True)
        wr = [w.clone().requires_grad_(True) for w in ws]
        br = [b.clone().requires_grad_(True) for b in bs]
        acts, inp = [], xr
        for i in range(L):
            inp = obf.linear(inp, wr[i], br[i])
            if i < L - 1:
                inp = torch.relu(inp)
            acts.append(inp)
        inp.backward(gy)
    # device
    wd = [w.to(dev) for w in ws]
    outs = [torch.empty(1, widths[i + 1], device=dev) for i in range(L)]
    ops.made_row_fwd(x.to(dev), [dict(w=wd[i], bias=bs[i].to(dev), relu=i < L - 1, out=outs[i]) for i in range(L)])
    for i in range(L):
        close(outs[i], acts[i].detach(), rtol=2e-3, atol_scale=2e-3, msg=f'y_{i}')
    gws = [torch.full((widths[i + 1], widths[i]), 9.0, device=dev) for i in range(L)]
    gbs = [torch.full((widths[i + 1],), 9.0, device=dev) for i in range(L)]
    gx = torch.empty(1, widths[0], device=dev)
    ops.made_row_bwd(gy.to(dev), [dict(w=wd[i], act=outs[i] if i < L - 1 else None, inp=outs[i - 1] if i > 0 else x.to(dev),
                                       gw=gws[i], gb=gbs[i]) for i in range(L)], g_x=gx)
    close(gx, xr.grad, rtol=5e-3, atol_scale=5e-3, msg='g_x')
    for i in range(L):
        close(gws[i], wr[i].grad, rtol=5e-3, atol_scale=5e-3, msg=f'gW_{i}')
        close(gbs[i], br[i].grad, rtol=5e-3, atol_scale=5e-3, msg=f'gb_{i}')
    # the all-zero input row of pass 0: x NULL, first layer's gw zero
    ops.made_row_fwd(None, [dict(w=wd[i], bias=bs[i].to(dev), relu=i < L - 1, out=outs[i]) for i in range(L)])
    close(outs[0], torch.relu(bs[0]).view(1, -1) if L > 1 else bs[0].view(1, -1), msg='zero row')
    ops.made_row_bwd(gy.to(dev), [dict(w=wd[i], act=outs[i] if i < L - 1 else None, inp=outs[i - 1] if i > 0 else None,
                                       gw=gws[i]) for i in range(L)])
    assert float(gws[0].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('n,d', [(130, 200), (64, 24), (77, 6), (1, 8)])
def test_iaf_update_bf16_kernels_match_the_formulas(ops, n, d):
    """gv_iaf_update_fwd_bf16 / _bwd_bf16 (the 4-column form when d % 4 == 0, the scalar form otherwise) against the update
    written out in torch (kgvae/flow_network.py:93-96 and its derivative): fp32 outputs, the bf16 row-major copies and the
    bf16 transposed copies."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(n * 7 + d)
    z = torch.randn(n, d, generator=g).to(dev)
    net = (torch.randn(n, 2 * d, generator=g) * 0.3).to(dev)
    xold = torch.randn(n, d, generator=g).to(dev)
    cnt = torch.randint(0, 3, (d,), generator=g).to(torch.int32).to(dev)
    npad = (n + 7) // 8 * 8
    dp = (d + 7) // 8 * 8
    xnew = torch.empty(n, d, device=dev)
    xb = torch.zeros(n, dp, dtype=torch.bfloat16, device=dev)
    xt = torch.zeros(d, npad, dtype=torch.bfloat16, device=dev)
    lib.call('gv_iaf_update_fwd_bf16', ptr(z), ptr(net), 2 * d, ptr(xold), ptr(cnt), ptr(xnew), ptr(xb), dp, ptr(xt), npad, n, d,
             lib.stream())
    act = (cnt > 0).view(1, -1)
    want = torch.where(act, z * torch.exp(net[:, d:] + net[:, :d]), xold)
    close(xnew, want, rtol=1e-5, atol_scale=1e-6, msg='x_new')
    assert torch.equal(xb[:, :d].float(), xnew.to(torch.bfloat16).float())
    assert torch.equal(xt[:, :n].float(), xb[:, :d].float().t()) and float(xt[:, n:].abs().max() if npad > n else 0.0) == 0.0
    # backward
    gx = torch.randn(n, d, generator=g).to(dev)
    gld = torch.randn(n, generator=g).to(dev)
    gz = torch.full((n, d), 0.5, device=dev)
    gnb = torch.zeros(n, 2 * dp, dtype=torch.bfloat16, device=dev)
    gnt = torch.zeros(2 * d, npad, dtype=torch.bfloat16, device=dev)
    gxold = torch.empty(n, d, device=dev)
    lib.call('gv_iaf_update_bwd_bf16', ptr(z), ptr(net), 2 * d, ptr(cnt), ptr(gx), ptr(gld), ptr(gz), ptr(gnb), 2 * dp, ptr(gnt),
             npad, ptr(gxold), n, d, lib.stream())
    ex = torch.exp(net[:, d:] + net[:, :d])
    gc = gx * cnt.view(1, -1).float()
    w_gz = torch.where(act, gc * ex, torch.zeros((), device=dev))
    w_gmu = torch.where(act, gc * z * ex, torch.zeros((), device=dev))
    w_gal = gld.view(-1, 1) + w_gmu
    close(gz, 0.5 + w_gz, rtol=1e-5, atol_scale=1e-6, msg='g_z accumulates')
    close(gxold, torch.where(act, torch.zeros((), device=dev), gx), msg='g_xold')
    close(gnb[:, :d].float(), w_gmu, rtol=1e-2, atol_scale=1e-2, msg='g_mu (bf16)')
    close(gnb[:, d:2 * d].float(), w_gal, rtol=1e-2, atol_scale=1e-2, msg='g_alpha (bf16)')
    assert torch.equal(gnt[:d, :n].float(), gnb[:, :d].float().t()) and torch.equal(gnt[d:, :n].float(), gnb[:, d:2 * d].float().t())


@pytest.mark.gpu
@pytest.mark.parametrize('m,n,k,split', [(200, 200, 4096, 8), (400, 200, 5000, 12), (72, 40, 1032, 3), (230, 448, 2048, 5)])
def test_gemm_bf16_gradw_weight_gradient_with_bias_gradient_from_the_same_pass(ops, m, n, k, split):
    """gv_gemm_bf16_gradw: c += A B^T over a long reduction on the whole-output kernel, a_rowsum += row sums of A -- against
    fp32 matmul / sums of the bf16 operands; both outputs ACCUMULATE."""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(dev).to(torch.bfloat16)
    b = torch.randn(n, k, generator=g).to(dev).to(torch.bfloat16)
    assert ops.gemm_bf16_gradw_fits(m, n, k, split)
    c = torch.full((m, n), 2.0, device=dev)
    rs = torch.full((m,), -1.0, device=dev)
    ops.gemm_bf16_gradw(a, b, m, n, k, c, accumulate=True, a_rowsum=rs, split_k=split)
    close(c, a.float().cpu() @ b.float().cpu().t() + 2.0, rtol=1e-4, atol_scale=1e-5, msg='c')
    close(rs, a.float().cpu().sum(dim=1) - 1.0, rtol=1e-4, atol_scale=1e-5, msg='row sums')
    c2 = torch.zeros(m, n, device=dev)
    ops.gemm_bf16_gradw(a, b, m, n, k, c2, accumulate=False, a_rowsum=None, split_k=split)
    close(c2, a.float().cpu() @ b.float().cpu().t(), rtol=1e-4, atol_scale=1e-5, msg='c, no row sums')
    assert not ops.gemm_bf16_gradw_fits(m, 1000, k, split) and not ops.gemm_bf16_gradw_fits(m, n, 64, split)


@pytest.mark.gpu
def test_multi_tensor_entry_points_of_the_made_path(ops):
    """gv_mul_multi, gv_made_pack_weight_multi and gv_rowsum_bf16_segments (several tensors per launch, host pointer tables)
    against their one-tensor counterparts."""
    import ctypes
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7)
    shapes = [(200, 200), (400, 200), (8, 24), (1, 1)]
    a = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    b = [torch.randn(*s, generator=g).to(dev) for s in shapes]
    # products: new outputs, and written into given tensors
    outs = ops.mul_multi(a, b)
    for o, x, y in zip(outs, a, b):
        assert torch.equal(o, x * y)
    given = [torch.full(s, 3.0, device=dev) for s in shapes]
    res = ops.mul_multi(a, b, outs=[given[0], None, given[2], None])
    assert res[0] is given[0] and res[2] is given[2] and torch.equal(given[0], a[0] * b[0]) and torch.equal(res[1], a[1] * b[1])
    # packing: every layer in one launch == one launch per layer
    ws = [x for x in a[:3]]
    multi = ops.made_pack_weights(ws)
    for w, (pf, pb) in zip(ws, multi):
        sf, sb = ops.made_pack_weight(w)
        assert torch.equal(pf.view(torch.int16), sf.view(torch.int16)) and torch.equal(pb.view(torch.int16), sb.view(torch.int16))
    # row sums of stacked bf16 rows, cut into segments with their own (accumulating) outputs; a NULL output skips its segment
    rows, cols, segs = 37, 9000, [5, 20, 12]
    x = torch.randn(rows, cols, generator=g).to(dev).to(torch.bfloat16)
    o = [torch.full((s,), 0.5, device=dev) for s in segs]
    tab = (ctypes.c_void_p * 3)(ptr(o[0]), None, ptr(o[2]))
    sg = (ctypes.c_int32 * 3)(*segs)
    wsp = torch.empty(int(lib.load().gv_rowsum_bf16_workspace_floats(rows, cols)), device=dev)
    lib.call('gv_rowsum_bf16_segments', ptr(x), x.stride(0), rows, cols, 3, ctypes.addressof(tab), ctypes.addressof(sg), 1, ptr(wsp),
             lib.stream())
    want = x.float().sum(dim=1)
    close(o[0], 0.5 + want[:5], rtol=1e-4, atol_scale=1e-5, msg='segment 0')
    close(o[2], 0.5 + want[25:], rtol=1e-4, atol_scale=1e-5, msg='segment 2')
    assert torch.equal(o[1], torch.full((20,), 0.5, device=dev))
