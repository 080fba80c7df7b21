"""K4 fused in fp32 (csrc/k_chain32.hip, gv_made_chain_f32): one launch per MADE pass on the fp32 MFMA that walks only the non-zero
groups of the masked weights (kgvae/flow_network.py:65-98).  Held to the launch-per-product path (gv_gemm_f32) BIT FOR BIT -- the
skipped terms are exact zeros -- and, through the whole MADE node, to the golden vectors generated from the reference
(tests/golden/made.npz, in test_gpu_model.py).   pytest -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _made(d, h, n_hidden, seed=3):
    from gcn_vae_amd.flows import MADE
    torch.manual_seed(seed)
    m = MADE(d, h, n_hidden).cuda()
    with torch.no_grad():          # biases away from zero so that ReLU patterns are not trivial
        for lin in m._linears():
            lin.bias.uniform_(-0.3, 0.3)
    return m


def _plan_words(plan):
    from gcn_vae_amd import made
    w = plan.cpu().numpy().astype(np.int64)
    assert w.size == made.PLAN_WORDS
    counts, lists = w[:4], w[4:4 + 4 * 256].reshape(4, 256)
    sets = w[4 + 4 * 256:].reshape(8, 16, 2)
    sets = (sets[..., 0] & 0xffffffff) | ((sets[..., 1] & 0xffffffff) << 32)
    return counts, lists, sets


@pytest.mark.parametrize('d,h,n_hidden', [(200, 200, 3), (40, 56, 2), (16, 16, 3)])
def test_plan_lists_every_tile_once_and_its_group_sets_are_the_masks_nonzero_groups(d, h, n_hidden):
    """gv_made_chain_f32_plan against a numpy reading of the same 0/1 masks: the group set of every (layer, 32-column tile), forward
    (B = W^T) and backward-x (B = W); every tile of every layer in exactly one wave's list, layers in ascending order; and the
    lower-triangular masks of create_masks really do leave groups out (that is the point)."""
    from gcn_vae_amd import made
    m = _made(d, h, n_hidden)
    masks = [l.mask for l in m._linears()]
    widths, kin = [k.shape[0] for k in masks], [k.shape[1] for k in masks]
    for transposed in (False, True):
        if transposed:
            ns, ks, ms = list(reversed(kin)), list(reversed(widths)), list(reversed(masks))
        else:
            ns, ks, ms = widths, kin, masks
        plan = made.made_chain_f32_plan(ns, ks, ms, transposed=transposed)
        torch.cuda.synchronize()
        counts, lists, sets = _plan_words(plan)
        seen = set()
        for w in range(4):
            units = [(int(e) >> 8, int(e) & 0x7f) for e in lists[w, :counts[w]]]          # (bit 7: the row half of a GV_C32_FINE build's units)
            assert all(not (int(e) & 0x80) for e in lists[w, :counts[w]])
            assert units == sorted(units, key=lambda u: u[0]) or all(units[i][0] <= units[i + 1][0] for i in range(len(units) - 1))
            for u in units:
                assert u not in seen
                seen.add(u)
        assert seen == {(l, t) for l in range(len(ns)) for t in range((ns[l] + 31) // 32)}
        skipped = total = 0
        for l, (n_, k_, mk) in enumerate(zip(ns, ks, ms)):
            b = mk.cpu().numpy().T if not transposed else mk.cpu().numpy()          # B [k][n]
            assert b.shape == (k_, n_)
            for t in range((n_ + 31) // 32):
                want = 0
                for g in range((k_ + 7) // 8):
                    if b[8 * g:8 * g + 8, 32 * t:32 * t + 32].any():
                        want |= 1 << g
                total += (k_ + 7) // 8
                skipped += (k_ + 7) // 8 - bin(want).count('1')
                assert int(sets[l, t]) == (want or 1), (transposed, l, t, hex(int(sets[l, t])), hex(want))
        if d >= 200:          # (narrow layers: a 32-column tile spans most of the degrees)
            assert skipped > 0.2 * total, (skipped, total)
    dense = made.made_chain_f32_plan(widths, kin, None)
    _, _, sets = _plan_words(dense)
    assert all(int(sets[l, t]) == (1 << ((kin[l] + 7) // 8)) - 1 for l in range(len(widths)) for t in range((widths[l] + 31) // 32))


@pytest.mark.parametrize('d,h,n_hidden,rows', [(200, 200, 3, 1000), (200, 200, 3, 65), (40, 56, 2, 300), (16, 16, 3, 1), (64, 128, 1, 129)])
def test_forward_and_backward_chains_equal_the_gemm_products_bit_for_bit(d, h, n_hidden, rows):
    """One gv_made_chain_f32 launch against the same products as gv_gemm_f32 launches (bias, ReLU; backward: ReLU mask on the
    operand, accumulating last product): every stored activation and gradient identical in every bit, with the masks' plan
    (zero groups skipped) and with the dense plan."""
    from gcn_vae_amd import made, ops
    m = _made(d, h, n_hidden)
    lin = m._linears()
    L = len(lin)
    ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
    bs = [l.bias.detach() for l in lin]
    masks = [l.mask for l in lin]
    widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
    assert made.made_chain_f32_fits(widths, kin) and made.made_chain_f32_fits(list(reversed(kin)), list(reversed(widths)))
    gen = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, d, generator=gen).cuda()
    # reference: a launch per product
    want, inp = [], x
    for l in range(L):
        inp = ops.gemm(inp, ws[l], trans_b=True, bias=bs[l], act=ops.ACT_RELU if l < L - 1 else ops.ACT_NONE)
        want.append(inp)
    g_top = torch.randn(rows, widths[-1], generator=gen).cuda()
    g_old0 = torch.randn(rows, d, generator=gen).cuda()
    want_g, g = [None] * L, g_top
    for l in reversed(range(1, L)):
        g = ops.gemm(g, ws[l], a_relu_mask=want[l] if l < L - 1 else None)
        want_g[l - 1] = torch.where(want[l - 1] > 0, g, torch.zeros_like(g))          # what the chain stores: the MASKED gradient
    want_gx = g_old0.clone()
    ops.gemm(g, ws[0], out=want_gx, accumulate=True, a_relu_mask=want[0])
    packed = made.made_pack_weights_f32(ws)
    for use_masks in (True, False):
        plan_f = made.made_chain_f32_plan(widths, kin, masks if use_masks else None)
        plan_b = made.made_chain_f32_plan(list(reversed(kin)), list(reversed(widths)), list(reversed(masks)) if use_masks else None,
                                          transposed=True)
        acts = [torch.full((rows, widths[l]), float('nan'), device='cuda') for l in range(L)]
        made.made_chain_f32(x, rows, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=acts[l])
                                      for l in range(L)], plan_f)
        for l in range(L):
            assert torch.equal(acts[l], want[l]), (use_masks, 'forward layer', l, float((acts[l] - want[l]).abs().max()))
        grads = [torch.full((rows, widths[l]), float('nan'), device='cuda') for l in range(L - 1)]
        gx = g_old0.clone()
        made.made_chain_f32(g_top, rows,
                            [dict(w_packed=packed[l][1], n=kin[l], k=widths[l], mask=acts[l - 1], out_f32=grads[l - 1])
                             for l in reversed(range(1, L))] +
                            [dict(w_packed=packed[0][1], n=d, k=widths[0], out_f32=gx, accumulate=True)], plan_b)
        for l in range(L - 1):
            assert torch.equal(grads[l], want_g[l]), (use_masks, 'backward layer', l, float((grads[l] - want_g[l]).abs().max()))
        assert torch.equal(gx, want_gx), (use_masks, float((gx - want_gx).abs().max()))
    assert float(want_gx.abs().max()) > 0 and all(float(w_.abs().max()) > 0 for w_ in want)


@pytest.mark.parametrize('d,h,n_hidden,rows', [(200, 200, 3, 700), (40, 56, 2, 300)])
def test_made_node_with_chains_equals_the_node_with_a_launch_per_product(monkeypatch, d, h, n_hidden, rows):
    """The fp32 MADE node (made._MADEForward) with one chain launch per pass, and with ALL passes + their IAF updates in one launch
    per direction (gv_made_passes_f32), against GV_MADE_CHAIN_F32=0: x, log-det, dL/dz and every parameter gradient bit for bit --
    through plain autograd and with the gradients going straight into FlatAdam's arena."""
    from gcn_vae_amd import made
    from gcn_vae_amd.optim import FlatAdam
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(5)).cuda()
    tags, inner = [], made.made_chain_f32
    monkeypatch.setattr(made, 'made_chain_f32', lambda *a, **k: (tags.append(k.get('tag')), inner(*a, **k))[1])
    for with_opt in (False, True):
        res = []
        for on, passes, gradw in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
            monkeypatch.setattr(made, 'MADE_CHAIN_F32', on)
            monkeypatch.setattr(made, 'MADE_PASSES_F32', passes)     # all passes + the IAF updates in one launch per direction
            monkeypatch.setattr(made, 'MADE_GRADW_F32', gradw)       # (its own summation order: held to the others within fp32 rounding)
            monkeypatch.setattr(made, 'MADE_ROW_F32', False)         # (pass 0 as one launch sums in another order too: its own test below)
            m = _made(d, h, n_hidden)
            opt = FlatAdam(list(m.parameters()), lr=1e-3, max_grad_norm=1.0) if with_opt else None
            if opt is not None:
                opt.zero_grad()
            zz = z.clone().requires_grad_(True)
            del tags[:]
            x, ld = m(zz)
            (x.sin().sum() + (ld * ld).sum()).backward()
            torch.cuda.synchronize()
            assert len(tags) == (2 * (len(m.m) - 1) if (on and not passes) else 0), (on, passes, tags)
            res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
            if opt is not None:
                opt.close()
        for other in res[1:]:
            assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1]) and torch.equal(res[0][2], other[2])
        assert torch.isfinite(res[0][2]).all() and float(res[0][2].abs().max()) > 0
        for a, b, b2, c in zip(res[0][3], res[1][3], res[2][3], res[3][3]):
            assert torch.equal(a, b) and torch.equal(a, b2) and float(a.abs().max()) > 0
            torch.testing.assert_close(c, a, rtol=2e-5, atol=2e-6 * float(a.abs().max()))
            assert bool(((a == 0) == (c == 0)).all()) or a.dim() == 1          # the masked-out entries are exact zeros in both


@pytest.mark.parametrize('out_f,in_f,rows', [(500, 500, 700), (1000, 500, 300), (72, 136, 129), (64, 16, 64)])
def test_products_that_skip_the_zero_blocks_of_a_mask_equal_the_dense_ones(out_f, in_f, rows):
    """gv_gemm_f32_sparse (ops.gemm(b_k_chunks= / c_tiles=)) with the block words of an autoregressive mask (ops.block_words) against
    the dense products of the masked weights: forward, backward-x and the weight gradient bit for bit (a skipped block only ever
    added exact zeros); with a device row count the padding tiles are skipped as in the dense entry."""
    from gcn_vae_amd import ops
    gen = torch.Generator().manual_seed(out_f + in_f)
    deg_out, deg_in = torch.arange(out_f) % max(in_f - 1, 1), torch.arange(in_f) % max(in_f - 1, 1)
    mask = (deg_out[:, None] >= deg_in[None, :]).float().cuda()
    w = (torch.randn(out_f, in_f, generator=gen).cuda() * mask).contiguous()
    x = torch.randn(rows, in_f, generator=gen).cuda()
    g = torch.randn(rows, out_f, generator=gen).cuda()
    bias = torch.randn(out_f, generator=gen).cuda()
    fwd, bwd, tiles = (ops.block_words(mask, k) for k in ('fwd', 'bwd', 'tiles'))
    zero_share = 1.0 - float(sum(bin(int(v) & (2 ** 64 - 1)).count('1') for v in fwd.tolist())) / (fwd.numel() * ((in_f + 15) // 16))
    assert zero_share > 0.2 or out_f < 128
    y0 = ops.gemm(x, w, trans_b=True, bias=bias, act=ops.ACT_RELU)
    y1 = ops.gemm(x, w, trans_b=True, bias=bias, act=ops.ACT_RELU, b_k_chunks=fwd)
    assert torch.equal(y0, y1) and float(y0.abs().max()) > 0
    gx0 = ops.gemm(g, w, a_relu_mask=y0)
    gx1 = ops.gemm(g, w, a_relu_mask=y0, b_k_chunks=bwd)
    assert torch.equal(gx0, gx1)
    acc0, acc1 = gx0.clone(), gx0.clone()
    ops.gemm(g, w, out=acc0, accumulate=True)
    ops.gemm(g, w, out=acc1, accumulate=True, b_k_chunks=bwd)
    assert torch.equal(acc0, acc1)
    for split in (1, 4):
        gw0 = ops.gemm(g, x, trans_a=True, split_k=split) * mask
        gw1 = ops.gemm(g, x, trans_a=True, split_k=split, c_tiles=tiles)
        assert torch.equal(gw0, gw1 * mask)
        assert bool((gw1[mask == 0].abs() < 3.4e38).all())                  # unwanted tiles hold zeros, not garbage
        t = mask.new_zeros(((out_f + 63) // 64) * 64, ((in_f + 63) // 64) * 64)
        t[:out_f, :in_f] = mask
        dead = ~(t.reshape(t.shape[0] // 64, 64, t.shape[1] // 64, 64).amax(dim=(1, 3)) > 0)
        full = dead.repeat_interleave(64, 0).repeat_interleave(64, 1)[:out_f, :in_f]
        assert float(gw1[full].abs().max() if bool(full.any()) else 0.0) == 0.0
    live = torch.tensor([rows // 2], dtype=torch.int32, device='cuda')
    with ops.live_rows(live, rows):
        a = ops.gemm(x, w, trans_b=True, bias=bias)
        b = ops.gemm(x, w, trans_b=True, bias=bias, b_k_chunks=fwd)
    assert torch.equal(a, b)


def test_made_node_per_layer_products_skip_the_masks_zero_blocks(monkeypatch):
    """A MADE too wide for the one-launch chain (per-layer products): with GV_MADE_SPARSE_F32 the products walk only the non-zero blocks
    of the masked weights -- x, log-det, dL/dz and every parameter gradient equal the dense run bit for bit."""
    from gcn_vae_amd import made
    d, h, rows = 72, 136, 333
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(9)).cuda()
    res, calls = [], []
    inner = made.gemm
    monkeypatch.setattr(made, 'gemm', lambda *a, **k: (calls.append(k.get('b_k_chunks') is not None or k.get('c_tiles') is not None), inner(*a, **k))[1])
    for sparse in (False, True):
        monkeypatch.setattr(made, 'MADE_CHAIN_F32', False)
        monkeypatch.setattr(made, 'MADE_SPARSE_F32', sparse)
        monkeypatch.setattr(made, 'MADE_ROW_F32', False)
        made._sparse_words.clear()
        m = _made(d, h, 2)
        zz = z.clone().requires_grad_(True)
        del calls[:]
        x, ld = m(zz)
        (x.sin().sum() + (ld * ld).sum()).backward()
        torch.cuda.synchronize()
        assert any(calls) == sparse
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_chain_skips_the_workgroups_that_hold_only_padding_rows():
    """ops.live_rows: a chain over a node array of exactly ``cap`` rows stores zeros for the 64-row workgroups past the device row
    count (nothing where it would accumulate) and computes the others in full."""
    from gcn_vae_amd import made, ops
    d, h, rows, live = 40, 56, 400, 150
    m = _made(d, h, 2)
    lin = m._linears()
    L = len(lin)
    ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
    bs = [l.bias.detach() for l in lin]
    widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
    x = torch.randn(rows, d, generator=torch.Generator().manual_seed(1)).cuda()
    packed = made.made_pack_weights_f32(ws)
    plan = made.made_chain_f32_plan(widths, kin, [l.mask for l in lin])
    full = [torch.empty(rows, widths[l], device='cuda') for l in range(L)]
    made.made_chain_f32(x, rows, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=full[l])
                                  for l in range(L)], plan)
    part = [torch.full((rows, widths[l]), 7.0, device='cuda') for l in range(L)]
    rows_dev = torch.tensor([live], dtype=torch.int32, device='cuda')
    with ops.live_rows(rows_dev, rows):
        made.made_chain_f32(x, rows, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=part[l],
                                           accumulate=(l == L - 1)) for l in range(L)], plan)
    edge = (live + 63) // 64 * 64
    for l in range(L - 1):
        assert torch.equal(part[l][:edge], full[l][:edge]) and float(part[l][edge:].abs().max()) == 0.0
    assert torch.equal(part[L - 1][:edge], full[L - 1][:edge] + 7.0) and bool((part[L - 1][edge:] == 7.0).all())


@pytest.mark.parametrize('m,n,k,masked', [(200, 200, 5000, True), (400, 200, 3000, True), (56, 40, 777, True), (200, 200, 64, False), (448, 256, 2100, False)])
def test_gradw_f32_against_a_double_precision_product(m, n, k, masked):
    """gv_made_gradw_f32: dW = wmask * (g^T a + g0m^T a0), db = column sums of g + g0m, against torch in fp64 -- storing and
    accumulating, with and without pass 0's row, under a MADE mask (tiles without a non-zero are skipped: exact zeros) and dense."""
    from gcn_vae_amd import made
    gen = torch.Generator().manual_seed(m + n + k)
    g = torch.randn(k, m, generator=gen).cuda()
    a = torch.randn(k, n, generator=gen).cuda()
    g0, act0, a0 = torch.randn(1, m, generator=gen).cuda(), torch.randn(1, m, generator=gen).cuda(), torch.randn(1, n, generator=gen).cuda()
    wmask = None
    if masked:
        deg_out, deg_in = torch.arange(m) % max(n - 1, 1), torch.arange(n) % max(n - 1, 1)
        wmask = (deg_out.unsqueeze(-1) >= deg_in.unsqueeze(0)).float().cuda()
    g0m = torch.where(act0 > 0, g0, torch.zeros_like(g0)).double()
    want = g.double().t() @ a.double() + g0m.t() @ a0.double()
    want_db = g.double().sum(0) + g0m.view(-1)
    if wmask is not None:
        want = want * wmask.double()
    scale = float(want.abs().max())
    out, db = made.made_gradw_f32(g, a, wmask=wmask, g0=g0, g0_act=act0, a0=a0)
    torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=2e-6 * scale)
    torch.testing.assert_close(db.double(), want_db, rtol=1e-5, atol=2e-6 * float(want_db.abs().max()))
    if wmask is not None:
        assert bool((out[wmask == 0] == 0).all())
    # accumulate into existing contents, no pass-0 row, no bias
    base = torch.randn(m, n, generator=gen).cuda()
    out2 = base.clone()
    made.made_gradw_f32(g, a, wmask=wmask, out=out2, accumulate=True, want_db=False)
    want2 = base.double() + (g.double().t() @ a.double()) * (wmask.double() if wmask is not None else 1.0)
    torch.testing.assert_close(out2.double(), want2, rtol=1e-5, atol=2e-6 * scale)
    db3 = torch.full((m,), 3.0, device='cuda')
    made.made_gradw_f32(g, a, wmask=wmask, db=db3, db_accumulate=True)
    torch.testing.assert_close(db3.double(), 3.0 + g.double().sum(0), rtol=1e-5, atol=2e-6 * float(want_db.abs().max()))


def test_gradw_f32_of_several_layers_in_one_launch_pair_is_the_single_products_bit_for_bit():
    """gv_made_gradw_f32_multi: the products of a MADE's layers (different shapes, masks, row terms, store / accumulate targets) in one
    launch pair = each of them through gv_made_gradw_f32."""
    from gcn_vae_amd import made
    gen = torch.Generator().manual_seed(77)
    k = 2900
    shapes = [(200, 40), (200, 200), (56, 200), (400, 200), (80, 56)]
    items, singles = [], []
    for q, (m, n) in enumerate(shapes):
        g, a = torch.randn(k, m, generator=gen).cuda(), torch.randn(k, n, generator=gen).cuda()
        wmask = None if q == 2 else ((torch.arange(m) % max(n - 1, 1)).unsqueeze(-1) >= (torch.arange(n) % max(n - 1, 1)).unsqueeze(0)).float().cuda()
        g0, act0, a0 = (torch.randn(1, m, generator=gen).cuda(), torch.randn(1, m, generator=gen).cuda(), torch.randn(1, n, generator=gen).cuda())
        base, bias = torch.randn(m, n, generator=gen).cuda(), torch.randn(m, generator=gen).cuda()
        kw = dict(wmask=wmask, g0=g0 if q != 1 else None, g0_act=act0 if q % 2 == 0 else None, a0=a0 if q != 1 else None,
                  accumulate=q == 3, db_accumulate=q in (0, 3), want_db=q != 4)
        singles.append(made.made_gradw_f32(g, a, out=base.clone(), db=bias.clone() if q != 4 else None, **kw))
        items.append(dict(g=g, a=a, out=base.clone(), db=bias.clone() if q != 4 else None, **kw))
    both = made.made_gradw_f32_multi(items)
    for (o1, b1), (o2, b2) in zip(singles, both):
        assert torch.equal(o1, o2)
        assert (b1 is None and b2 is None) or torch.equal(b1, b2)


@pytest.mark.parametrize('d,h,n_hidden,rows', [(200, 200, 3, 500), (40, 56, 2, 300)])
def test_made_node_with_pass_0_as_single_workgroup_launches(monkeypatch, d, h, n_hidden, rows):
    """GV_MADE_ROW_F32: pass 0 of the fp32 node (one broadcast row) on gv_made_row_fwd / _bwd with exact fp32 operands instead of
    one-row products on the GEMM: same node within fp32 rounding (the row kernels sum along k in another order)."""
    from gcn_vae_amd import made
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(9)).cuda()
    res = []
    for on in (False, True):
        monkeypatch.setattr(made, 'MADE_ROW_F32', on)
        m = _made(d, h, n_hidden)
        zz = z.clone().requires_grad_(True)
        x, ld = m(zz)
        (x.sin().sum() + (ld * ld).sum()).backward()
        torch.cuda.synchronize()
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    for a, b in zip(res[0][:3], res[1][:3]):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-6 * float(a.abs().max()))
    for a, b in zip(res[0][3], res[1][3]):
        torch.testing.assert_close(b, a, rtol=5e-5, atol=5e-6 * float(a.abs().max()))


def test_fused_passes_on_a_padded_batch_equal_the_launch_per_pass_path(monkeypatch):
    """ops.live_rows with the fused passes: workgroups that hold only padding rows skip the layers (zeros out) but still run the
    IAF updates of their rows, exactly as the per-pass launches + update kernels do -- whole node, bit for bit, forward and backward."""
    from gcn_vae_amd import made, ops
    d, h, rows, live = 40, 56, 400, 150
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(2)).cuda()
    rows_dev = torch.tensor([live], dtype=torch.int32, device='cuda')
    res = []
    for passes in (False, True):
        monkeypatch.setattr(made, 'MADE_PASSES_F32', passes)
        monkeypatch.setattr(made, 'MADE_GRADW_F32', False)
        monkeypatch.setattr(made, 'MADE_ROW_F32', False)
        m = _made(d, h, 2)
        zz = z.clone().requires_grad_(True)
        with ops.live_rows(rows_dev, rows):
            x, ld = m(zz)
            w = torch.zeros(rows, 1, device='cuda')
            w[:live] = 1.0                      # the padding rows take no part in the loss
            ((x * w).sin().sum() + ((ld * w.view(-1)) ** 2).sum()).backward()
        torch.cuda.synchronize()
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_non_finite_input_in_a_masked_out_group_is_skipped_here_and_poisons_the_reference(monkeypatch):
    """Documented difference.  The reference forms mask * weight and multiplies EVERYTHING
    (kgvae/flow_network.py:14-15): an infinite activation times a masked (zero) weight is NaN, torch.relu hands the NaN on, and the
    whole row of the node's output is NaN.  Here (a) the fp32 chain multiplies only the 8-deep groups in which the mask holds a
    non-zero, so a unit whose mask excludes the whole group of the infinite input never sees it, and (b) the kernels' ReLU is
    fmaxf(x, 0), which returns 0 for a NaN: what a dense plan (GV_MADE_CHAIN_F32_SKIP=0) does produce is scrubbed at the next
    activation.  The row stays finite except for the infinite column itself; its NaNs are a subset of the reference's; the other
    rows are untouched.  On finite inputs the node is bit-identical to the launch-per-product path (the tests above)."""
    from gcn_vae_amd import made
    from oracle import flows as oflows
    d, h, rows = 200, 200, 64
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(21))
    z[0, d - 2] = float('inf')          # degree d - 2: masked out of every hidden unit but one
    outs = {}
    for skip in (True, False):
        monkeypatch.setattr(made, 'MADE_CHAIN_F32_SKIP', skip)
        made._chain32_plans.clear()
        m = _made(d, h, 3)
        with torch.no_grad():
            x, ld = m(z.cuda())
        torch.cuda.synchronize()
        outs[skip] = x.cpu()
        state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    made._chain32_plans.clear()
    xo, _ = oflows.made_forward(z, oflows.made_layers_from_state(state, '', 3), d, h, 3)
    assert bool(torch.isnan(xo[0]).all()), 'the reference poisons the whole row'
    for skip, x in outs.items():
        assert not torch.isnan(x[0]).any(), skip                    # (a subset of the reference's NaNs: here the empty one)
        assert int(torch.isinf(x[0]).sum()) == 1 and bool(torch.isinf(x[0, d - 2])), skip
        torch.testing.assert_close(x[1:], xo[1:], rtol=1e-4, atol=1e-5)          # the other rows: the reference's values


def test_padding_rows_with_poisoned_gradients_do_not_reach_the_parameter_gradients():
    """ops.live_rows + the fused passes + gv_made_gradw_f32: the weight / bias gradient products reduce over all stacked rows, padding
    included, so the update's backward stores ZEROS as [g_mu | g_alpha] of padding rows whatever dL/dx and dL/dlogdet hold there (a
    loss that does not mask them: here NaN-poisoned).  Parameter gradients = the node on the live rows alone; dL/dz of the live rows too."""
    from gcn_vae_amd import ops
    d, h, rows, live = 40, 56, 400, 150
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(4)).cuda()
    rows_dev = torch.tensor([live], dtype=torch.int32, device='cuda')
    probe = torch.randn(rows, d, generator=torch.Generator().manual_seed(5)).cuda()
    ref_m = _made(d, h, 2)
    zr = z[:live].clone().requires_grad_(True)
    x, ld = ref_m(zr)
    ((x * probe[:live]).sum() + (ld ** 2).sum()).backward()
    m = _made(d, h, 2)
    zz = z.clone().requires_grad_(True)
    with ops.live_rows(rows_dev, rows):
        x2, ld2 = m(zz)
        gx = probe.clone()
        gx[live:] = float('nan')
        gl = 2 * ld2.detach().clone()
        gl[live:] = float('nan')
        torch.autograd.backward([x2, ld2], [gx, gl])
    torch.cuda.synchronize()
    torch.testing.assert_close(zz.grad[:live], zr.grad, rtol=2e-5, atol=2e-6 * float(zr.grad.abs().max()))
    for a, b in zip(ref_m.parameters(), m.parameters()):
        assert torch.isfinite(b.grad).all()
        torch.testing.assert_close(b.grad, a.grad, rtol=5e-5, atol=5e-6 * float(a.grad.abs().max()))


@pytest.mark.parametrize('passes', [True, False])
def test_made_block_with_the_permute_layer_folded_in_equals_block_then_permute(monkeypatch, passes):
    """MADE.forward(z, reverse_out=True) -- the PermuteLayer behind an IAF block as reversed column stores of the block's last
    launch, reversed reads of dL/dx in its backward launch -- against the block followed by the layer: bit for bit, values and
    gradients; with the launch-per-pass path (explicit gv_reverse_cols around the node) as well."""
    from gcn_vae_amd import flows, made
    monkeypatch.setattr(made, 'MADE_PASSES_F32', passes)
    d, h, rows = 200, 200, 777
    z = torch.randn(rows, d, generator=torch.Generator().manual_seed(12)).cuda()
    probe = torch.randn(rows, d, generator=torch.Generator().manual_seed(13)).cuda()
    perm = flows.PermuteLayer(d)
    res = []
    for fold in (False, True):
        m = _made(d, h, 3)
        zz = z.clone().requires_grad_(True)
        if fold:
            x, ld = m(zz, reverse_out=True)
        else:
            x, ld = m(zz)
            x = perm(x)[0]
        ((x * probe).sum() + (ld ** 2).sum()).backward()
        torch.cuda.synchronize()
        res.append((x.detach().clone(), ld.detach().clone(), zz.grad.clone(), [p.grad.detach().clone() for p in m.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


def test_mean_rows_multi_is_flow_log_prob_and_its_gradient():
    """ops.mean_rows_multi (gv_mean_rows_multi / _bwd) = mean over the rows that exist of the summed per-row log-determinants
    (kgvae/model.py:116-123), against torch: all rows, a row limit (rows riding along behind the first n), a device row count."""
    from gcn_vae_amd import ops
    gen = torch.Generator().manual_seed(4)
    n, extra, live = 5000, 200, 3777
    xs = [torch.randn(n + extra, generator=gen).cuda().requires_grad_(True) for _ in range(3)]
    rows_dev = torch.tensor([live], dtype=torch.int32, device='cuda')
    for limit, rd, count in ((None, None, n + extra), (n, None, n), (n, rows_dev, live)):
        for x in xs:
            x.grad = None
        out = ops.mean_rows_multi(xs, n=limit, rows_dev=rd)
        want = sum(x.detach().double()[:count] for x in xs).sum() / count
        torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=1e-6)
        (out * 3.0).backward()
        g = torch.zeros(n + extra, dtype=torch.float64, device='cuda')
        g[:count] = 3.0 / count
        for x in xs:
            torch.testing.assert_close(x.grad.double(), g, rtol=1e-6, atol=0.0)


def test_accumulating_update_backward_equals_the_two_launch_form():
    """gv_iaf_update_bwd_acc (four columns per thread, dL/dz added in place or written) against gv_iaf_update_bwd + an add: every
    output bit for bit, with and without a log-det gradient, with columns that are handed through (count 0) and counted twice."""
    from gcn_vae_amd import lib
    from gcn_vae_amd.lib import ptr
    gen = torch.Generator().manual_seed(8)
    n, d = 1000, 40
    z, net, gx = (torch.randn(n, w, generator=gen).cuda() for w in (d, 2 * d, d))
    net = net * 0.3
    gld = torch.randn(n, generator=gen).cuda()
    cnt = torch.ones(d, dtype=torch.int32, device='cuda')
    cnt[-1], cnt[0] = 0, 2
    for use_gld in (True, False):
        gz_ref, gnet_ref, gold_ref = (torch.empty(n, w, device='cuda') for w in (d, 2 * d, d))
        lib.call('gv_iaf_update_bwd', ptr(z), ptr(net), 2 * d, ptr(cnt), ptr(gx), ptr(gld) if use_gld else None, ptr(gz_ref), ptr(gnet_ref),
                 ptr(gold_ref), n, d, lib.stream())
        base = torch.randn(n, d, generator=gen).cuda()
        for accumulate in (0, 1):
            gz = base.clone()
            gnet, gold = torch.empty(n, 2 * d, device='cuda'), torch.empty(n, d, device='cuda')
            lib.call('gv_iaf_update_bwd_acc', ptr(z), ptr(net), 2 * d, ptr(cnt), ptr(gx), ptr(gld) if use_gld else None, ptr(gz), accumulate,
                     ptr(gnet), ptr(gold), n, d, lib.stream())
            assert torch.equal(gz, base + gz_ref if accumulate else gz_ref)
            assert torch.equal(gnet, gnet_ref) and torch.equal(gold, gold_ref)
    assert float(gold_ref[:, -1].abs().max()) > 0 and float(gold_ref[:, :-1].abs().max()) == 0


def test_row_kernels_with_exact_fp32_operands_equal_one_row_products():
    """gv_made_row_fwd / _bwd with layers[0].reserved = 1 (the fp32 node's pass 0): the masked MLP on ONE row and its backward with
    exact fp32 operands, against the same chain as one-row products on gv_gemm_f32 (another summation order along k: fp32 rounding)."""
    from gcn_vae_amd import made, ops
    m = _made(200, 200, 3)
    lin = m._linears()
    L = len(lin)
    ws = [ops.masked_weight(l.mask, l.weight).detach() for l in lin]
    bs = [l.bias.detach() for l in lin]
    acts = [torch.empty(1, w.shape[0], device='cuda') for w in ws]
    made.made_row_fwd(None, [dict(w=ws[l], bias=bs[l], relu=l < L - 1, out=acts[l]) for l in range(L)], exact=True)
    want, inp = [], torch.zeros(1, 200, device='cuda')
    for l in range(L):
        inp = ops.gemm(inp, ws[l], trans_b=True, bias=bs[l], act=ops.ACT_RELU if l < L - 1 else ops.ACT_NONE)
        want.append(inp)
    for a, b in zip(acts, want):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    g_top = torch.randn(1, ws[-1].shape[0], generator=torch.Generator().manual_seed(1)).cuda()
    gms = [torch.empty(1, w.shape[0], device='cuda') for w in ws]
    made.made_row_bwd(g_top, [dict(w=ws[l], act=acts[l] if l < L - 1 else None, gb=gms[l]) for l in range(L)], exact=True)
    g = g_top
    for l in reversed(range(L)):
        gm = g if l == L - 1 else torch.where(want[l] > 0, g, torch.zeros_like(g))
        torch.testing.assert_close(gms[l], gm, rtol=1e-5, atol=1e-6 * float(gm.abs().max() + 1e-30))
        if l > 0:
            g = ops.gemm(gm, ws[l])
