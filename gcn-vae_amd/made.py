"""K4: the masked MLP of the IAF blocks (kgvae/flow_network.py:62-98, MADE / MaskedLinear) as autograd nodes over the HIP entry
points of csrc/k_made.hip, k_chain.hip, k_chain32.hip, k_gradw32.hip, k_row.hip and k_gemm.hip.

  made_forward            MADE.forward as ONE node: _MADEForwardBF16 (bf16-stored operands, one gv_made_chain launch per pass, the
                          IAF update in the chain's last layer; BASELINE configs[2]) or _MADEForward (fp32, the reference's own
                          precision: all stacked passes + their IAF updates in one gv_made_passes_f32 launch per direction, weight
                          and bias gradients on gv_made_gradw_f32, pass 0 on the row kernels; launch-per-product fallback)
  made_chain_f32 / made_passes_f32 / made_gradw_f32 / made_pack_weights_f32 / made_*_plan   the fp32 K4 entry points (csrc/k_chain32.hip,
                          k_gradw32.hip): zero groups / tiles of the masked weights are not multiplied
  made_prepare            the parameter-only part of the announced calls (mask folds, packed weights, pass 0's row) on a side stream
  made_chain / made_row_* / made_pack_weight* / gemm_bf16_* / cast_bf16 / dense_bf16*   thin wrappers of the C-ABI entry points
  _by_row_blocks          a node's passes over independent row blocks on their own streams

Part of the ``ops`` namespace (ops re-exports everything here: callers keep writing ops.made_forward); the knobs of this file
(MADE_*, GRADW_SPLIT_MAX, ...) are THIS module's globals.
"""
import contextlib
import ctypes as _ct
import os as _os

import torch

from . import lib
from . import ops as _ops
from .lib import ptr
from .ops import (ACT_NONE, ACT_RELU, GRAD_FRESH, _chk, _direct, _row_major, _side, _stamp_direct, _verify_direct,
                  backward_side, colsum, gemm, iaf_bwd_row0, masked_weight, mul_multi, pick_split_k)


_zero_rows = {}


def _zero_row(d, device):
    """One read-only all-zero row (1, d) per device and width: pass 0's input (not a fill per call)."""
    key = (device, int(d))
    if key not in _zero_rows:
        row = torch.zeros(1, int(d), dtype=torch.float32, device=device)
        if _capturing():
            return row      # a tensor filled by a RECORDED launch lives in the graph's pool and is uninitialised until a replay: not cached
        _zero_rows[key] = row
    return _zero_rows[key]


def _capturing():
    """Is the current stream being captured?  The process-wide caches of this module (the zero row, the chain and weight-gradient
    plans) are filled by a launch on the current stream: under capture that launch is only recorded, so the tensor is built for
    that call alone and not cached -- an eager use before the first replay would read uninitialised memory."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


MADE_SPARSE_F32 = _os.environ.get('GV_MADE_SPARSE_F32', '1') == '1'      # per-layer products (widths outside the chain kernel) skip the masks' zero blocks
_sparse_words = {}


def _mask_words(masks):
    """Per layer the block words of its autoregressive mask for the three products of the per-layer path (ops.block_words: forward,
    backward-x, weight-gradient tiles); None where they do not apply.  Built once per mask set (a host read-back of static masks:
    never under capture -- a step captured before any eager use runs dense)."""
    if masks is None or not MADE_SPARSE_F32:
        return None
    key = tuple((m.data_ptr(), tuple(m.shape), m._version) for m in masks)
    hit = _sparse_words.get(key)
    if hit is None:
        if _capturing():
            return None
        if len(_sparse_words) > 64:
            _sparse_words.clear()
        hit = _sparse_words[key] = ([dict(fwd=_ops.block_words(m, 'fwd'), bwd=_ops.block_words(m, 'bwd'), tiles=_ops.block_words(m, 'tiles'))
                                     for m in masks], masks)       # (holds the masks: the key is their addresses)
    return hit[0]


def _made_params_work_f32(masks, ws, bs, d, S):
    """The part of _MADEForward.forward that depends on the parameters alone: the mask fold (one launch), pass 0 on its single zero
    row -- one single-workgroup launch with exact fp32 operands where the widths allow, else a launch per product --, and for the
    chain kernel (gv_made_chain_f32: activations stay in LDS between the layers, the zero groups of the masked weights are not
    multiplied; otherwise -- and with GV_MADE_CHAIN_F32=0 -- a launch per product) the fragment-packed weights and the unit plan."""
    L = len(ws)
    if masks is not None:
        ws = mul_multi(masks, ws)
    dev = ws[0].device
    f32 = dict(dtype=torch.float32, device=dev)
    zero_row = _zero_row(d, dev)
    row = (MADE_ROW_F32 and L <= 8 and d <= 512 and max(w.shape[0] for w in ws) <= 512 and all(w.shape[1] % 4 == 0 for w in ws))
    if row:
        acts0 = [torch.empty(1, w.shape[0], **f32) for w in ws]
        made_row_fwd(None, [dict(w=ws[l], bias=bs[l], relu=l < L - 1, out=acts0[l]) for l in range(L)], exact=True)
    else:
        acts0, inp = [], zero_row
        for l in range(L):
            inp = gemm(inp, ws[l], trans_b=True, bias=bs[l], act=ACT_RELU if l < L - 1 else ACT_NONE)
            acts0.append(inp)
    widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
    chain = (MADE_CHAIN_F32 and S > 0 and L <= 8 and d % 8 == 0 and made_chain_f32_fits(widths, kin)
             and made_chain_f32_fits(list(reversed(kin)), list(reversed(widths))))
    packed = plan_f = None
    if chain:
        packed = made_pack_weights_f32(ws)
        plan_f = made_chain_f32_plan(widths, kin, masks)
    return dict(f32=True, ws=ws, row=row, acts0=acts0, zero_row=zero_row, chain=chain, packed=packed, plan_f=plan_f)


class _MADEForward(torch.autograd.Function):
    """MADE.forward (kgvae/flow_network.py:85-98) as ONE autograd node.

    The reference runs ``len(self.m)`` (= n_hidden + 3) sequential passes of the masked MLP, each followed by
    ``x[:, i] = z[:, i] * exp(alpha[:, i] + mu[:, i])``.  Here
      * pass 0 -- whose input is the all-zero matrix, i.e. N identical rows -- evaluates the MLP on ONE row and
        broadcasts it into the update (its backward sums the update's gradient over the rows and back-propagates
        that single row);
      * every later pass writes its layer activations into row slices of per-layer buffers stacked over the passes,
        so the backward walks the passes in reverse with one NN GEMM per layer whose A operand is read through the
        ReLU mask of the stored activation (no masking kernels, no materialised masked gradient), and forms each
        layer's weight gradient ONCE as a single split-K product over all stacked rows (K = (passes-1) x N); the
        reference's autograd does it per pass and sums.
    Inputs: z (N, D); masked weights W_l (out_l, in_l) and biases, l = 0..L-1 (last layer: 2D outputs [mu | alpha]);
    colcount int32 (passes, D).  Outputs: x (N, D), log_det (N,) = sum_d alpha of the last pass.
    """

    @staticmethod
    def forward(ctx, z, colcount, masks, reverse_out, *wb):
        ctx.set_materialize_grads(False)
        L = len(wb) // 2
        ws, bs = wb[:L], wb[L:]
        ctx.reverse_out = bool(reverse_out)      # x comes out with its columns reversed (the PermuteLayer behind the block rides along)
        # masks: the autoregressive masks when the weights are the RAW parameters -- folded here in one launch, and in the backward
        # pass the masked weight gradients go straight into the optimiser's arena where that is registered (then, with the bias
        # gradients, on the side stream: ops.backward_side); None: the weights are masked already
        ctx.masks = masks
        ctx.direct_w = [_direct(w) for w in ws] if masks is not None else [None] * L
        ctx.direct_b = [_direct(b) if b is not None else None for b in bs]
        _stamp_direct(ctx)
        z = _chk(z.contiguous(), name='z')
        n, d = z.shape
        P = colcount.shape[0]
        f32 = dict(dtype=torch.float32, device=z.device)
        st = lib.stream()
        S = P - 1                                                 # passes 1..P-1 are stacked; pass 0 is one row
        # what depends on the parameters alone (mask fold, fragment-packed weights, pass 0's row): done ahead on a side stream where
        # the model announced this call (made_prepare), here otherwise
        prep = _made_prep.pop(_made_prep_key(ws, d, S), None)
        if prep is not None and prep.get('f32'):
            if not prep['joined'][0]:       # ONE join for everything that was prepared
                torch.cuda.current_stream().wait_event(prep['done'])
                prep['joined'][0] = True
        else:
            prep = _made_params_work_f32(masks, ws, bs, d, S)
        ws, row, acts0, zero_row, chain, packed, plan_f = (prep[k_] for k_ in ('ws', 'row', 'acts0', 'zero_row', 'chain', 'packed', 'plan_f'))
        widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
        xin = torch.empty(max(S, 1) * n, d, **f32)               # xin[s] = input of pass s+1 = output of pass s
        acts = [torch.empty(max(S, 1) * n, ws[l].shape[0], **f32) for l in range(L)]   # acts[L-1] = [mu | alpha]
        x_out = torch.empty(n, d, **f32)
        first_out = xin[0:n] if P > 1 else x_out
        # ONE launch for passes 1 .. P-1 and their IAF updates (below); pass 0's update, the log-determinant and the column reversal
        # of a PermuteLayer behind the block ride in it
        fused_passes = chain and MADE_PASSES_F32 and d % 4 == 0 and lib.TIMER is None
        log_det = torch.empty(n, **f32)
        # (columns outside the first index set would keep x_old = 0: flows.MADE checks at construction that there are none)
        if not fused_passes:
            lib.call('gv_iaf_update_fwd', ptr(z), ptr(acts0[L - 1]), 0, ptr(z), ptr(colcount[0]), ptr(first_out), n, d, st)
        words = None if chain else _mask_words(masks)       # zero blocks of the masked weights: skipped by the per-layer products
        ctx.words = words

        def passes(r0, r1):      # passes 1 .. P-1 for the rows [r0, r1): every launch of a pass is row-local
            for p in range(1, P):
                a, b = (p - 1) * n + r0, (p - 1) * n + r1
                inp = xin[a:b]
                if chain:
                    made_chain_f32(inp, r1 - r0, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1,
                                                       out_f32=acts[l][a:b]) for l in range(L)], plan_f, tag='madechain_fwd_f32')
                    inp = acts[L - 1][a:b]
                for l in range(L) if not chain else ():
                    out = acts[l][a:b]
                    gemm(inp, ws[l], trans_b=True, bias=bs[l], act=ACT_RELU if l < L - 1 else ACT_NONE, out=out,
                         b_k_chunks=None if words is None else words[l]['fwd'])
                    inp = out
                nxt = xin[p * n + r0:p * n + r1] if p + 1 < P else x_out[r0:r1]
                lib.call('gv_iaf_update_fwd', ptr(z[r0:r1]), ptr(inp), 2 * d, ptr(xin[a:b]), ptr(colcount[p]), ptr(nxt), r1 - r0, d,
                         lib.stream())
        if fused_passes:
            # everything a pass does is local to a workgroup's 64 rows, so the workgroup walks the passes itself (stacked buffers: n
            # rows per pass); 12 launches per MADE become 1
            made_passes_f32(xin, n, [dict(w_packed=packed[l][0], n=widths[l], k=kin[l], bias=bs[l], relu=l < L - 1, out_f32=acts[l][0:n])
                                     for l in range(L)], plan_f,
                            dict(mode=1, passes=S, step=n, d=d, z=z, colcount=colcount[1], x_out=x_out, net0=acts0[L - 1], cnt0=colcount[0],
                                 log_det=log_det, flags=1 | 4 | 8 | (16 if ctx.reverse_out else 0)), tag='madechain_fwd_f32')
        elif chain:       # MFMA-bound launches that fill the chip: one sequence over all rows (and the padding rows are skipped per workgroup)
            passes(0, n)
        else:
            _by_row_blocks(passes, n, MADE_F32_ROW_BLOCKS, MADE_F32_ROW_BLOCKS_MIN_TILES)
        if fused_passes:
            pass
        elif P > 1:
            lib.call('gv_rowsum', ptr(acts[L - 1][(S - 1) * n:]), 2 * d, d, d, ptr(log_det), n, st)
        else:
            log_det = acts0[L - 1][:, d:].sum(dim=1).expand(n).contiguous()
        if ctx.reverse_out and not fused_passes:
            rev = torch.empty_like(x_out)
            lib.call('gv_reverse_cols', ptr(x_out), ptr(rev), n, d, st)
            x_out = rev
        ctx.save_for_backward(z, colcount, xin, zero_row, *acts, *acts0, *ws, *([pk[1] for pk in packed] if chain else []))
        ctx.L = L
        ctx.chain = chain
        ctx.row = row
        ctx.has_bias = [b is not None for b in bs]
        return x_out, log_det

    @staticmethod
    def backward(ctx, gx, gld):
        L = ctx.L
        saved = ctx.saved_tensors
        z, colcount, xin, zero_row = saved[:4]
        acts, acts0, ws = saved[4:4 + L], saved[4 + L:4 + 2 * L], saved[4 + 2 * L:4 + 3 * L]
        chain, wpb = ctx.chain, saved[4 + 3 * L:4 + 4 * L]          # fragment-packed B = W of every layer (backward-x chain)
        n, d = z.shape
        P = colcount.shape[0]
        S = P - 1
        f32 = dict(dtype=torch.float32, device=z.device)
        st = lib.stream()
        widths, kin = [w.shape[0] for w in ws], [w.shape[1] for w in ws]
        plan_b = made_chain_f32_plan(list(reversed(kin)), list(reversed(widths)),
                                     list(reversed(ctx.masks)) if ctx.masks is not None else None, transposed=True) if chain else None
        gx = torch.zeros(n, d, **f32) if gx is None else _chk(gx.contiguous(), name='gx')
        gld = None if gld is None else _chk(gld.contiguous(), name='gld')
        fused_passes = chain and MADE_PASSES_F32 and d % 4 == 0 and P > 1 and lib.TIMER is None
        if ctx.reverse_out and not fused_passes:          # (the fused launch reads dL/dx with its columns reversed)
            rev = torch.empty_like(gx)
            lib.call('gv_reverse_cols', ptr(gx), ptr(rev), n, d, st)
            gx = rev
        grads = [torch.empty(max(S, 1) * n, ws[l].shape[0], **f32) for l in range(L)]   # grad w.r.t. each layer's OUTPUT
        acc_gz = d % 4 == 0 and P > 1                             # the update's backward adds dL/dz in place; the first pass run WRITES it
        g_z = torch.empty(n, d, **f32) if acc_gz else torch.zeros(n, d, **f32)
        gz_p = None if acc_gz else torch.empty(n, d, **f32)
        gold_stack = torch.empty(max(S, 1) * n, d, **f32)                 # dL/dx_old of every pass, stacked like the activations
        g_olds = {p: gold_stack[(p - 1) * n:p * n] for p in range(1, P)}

        words = getattr(ctx, 'words', None)

        def passes(r0, r1):      # the backward of passes P-1 .. 1 for the rows [r0, r1): every launch is row-local
            g_in, m = gx, r1 - r0
            for p in reversed(range(1, P)):
                a, b = (p - 1) * n + r0, (p - 1) * n + r1
                g_old = g_olds[p][r0:r1]
                if d % 4 == 0:       # dL/dz of the pass added to the running sum by the same launch
                    lib.call('gv_iaf_update_bwd_acc', ptr(z[r0:r1]), ptr(acts[L - 1][a:b]), 2 * d, ptr(colcount[p]), ptr(g_in[r0:r1]),
                             ptr(gld[r0:r1]) if (p == P - 1 and gld is not None) else None, ptr(g_z[r0:r1]), 0 if p == P - 1 else 1,
                             ptr(grads[L - 1][a:b]),
                             ptr(g_old), m, d, lib.stream())
                else:
                    lib.call('gv_iaf_update_bwd', ptr(z[r0:r1]), ptr(acts[L - 1][a:b]), 2 * d, ptr(colcount[p]), ptr(g_in[r0:r1]),
                             ptr(gld[r0:r1]) if (p == P - 1 and gld is not None) else None, ptr(gz_p[r0:r1]), ptr(grads[L - 1][a:b]),
                             ptr(g_old), m, d, lib.stream())
                    lib.call('gv_axpby', m * d, None, 1.0, ptr(gz_p[r0:r1]), 1.0, ptr(g_z[r0:r1]), lib.stream())
                if chain:       # g_{l-1} = (g_l W_l) * [a_{l-1} > 0] down to g_x, one launch; the hidden gradients are STORED masked
                    made_chain_f32(grads[L - 1][a:b], m,
                                   [dict(w_packed=wpb[l], n=kin[l], k=widths[l], mask=acts[l - 1][a:b], out_f32=grads[l - 1][a:b])
                                    for l in reversed(range(1, L))] +
                                   [dict(w_packed=wpb[0], n=d, k=widths[0], out_f32=g_old, accumulate=True)], plan_b,
                                   tag='madechain_bwd_f32')
                for l in reversed(range(L)) if not chain else ():
                    mask = acts[l][a:b] if l < L - 1 else None
                    kw = None if words is None else words[l]['bwd']
                    if l > 0:
                        gemm(grads[l][a:b], ws[l], out=grads[l - 1][a:b], a_relu_mask=mask, b_k_chunks=kw)
                    else:       # gradient w.r.t. the pass's input x_p joins the update's pass-through gradient
                        gemm(grads[0][a:b], ws[0], out=g_old, accumulate=True, a_relu_mask=mask, b_k_chunks=kw)
                g_in = g_olds[p]
        if fused_passes:
            # passes P-1 .. 1, each = the update's backward + the backward-x chain, in ONE launch (the buffers of consecutive passes
            # are n rows apart, walked downwards)
            a0 = (S - 1) * n
            made_passes_f32(grads[L - 1][a0:a0 + n], n,
                            [dict(w_packed=wpb[l], n=kin[l], k=widths[l], mask=acts[l - 1][a0:a0 + n], out_f32=grads[l - 1][a0:a0 + n])
                             for l in reversed(range(1, L))] +
                            [dict(w_packed=wpb[0], n=d, k=widths[0], out_f32=gold_stack[a0:a0 + n], accumulate=True)], plan_b,
                            dict(mode=2, passes=S, step=-n, d=d, z=z, colcount=colcount[P - 1], net=acts[L - 1][a0:a0 + n], g_in=gx,
                                 g_logdet=gld, g_z=g_z, flags=(1 if gld is not None else 0) | 2 | (4 if ctx.reverse_out else 0)),
                            tag='madechain_bwd_f32')
        elif chain:
            passes(0, n)
        else:
            _by_row_blocks(passes, n, MADE_F32_ROW_BLOCKS, MADE_F32_ROW_BLOCKS_MIN_TILES)
        g_cur = g_olds[1] if P > 1 else gx
        # pass 0: the update's gradient w.r.t. the broadcast net row is its column sum; x_old was the zero matrix
        g_row = iaf_bwd_row0(z, acts0[L - 1], colcount[0], g_cur, gld if P == 1 else None, g_z)      # (1, 2D)
        rows0 = [None] * L                                        # single-row gradients per layer output (masked where it is used)
        # chain path: the hidden gradients are stored ReLU-masked, so each layer's weight AND bias gradient over all stacked passes,
        # pass 0's rank-1 term, the mask fold and the store / add into the arena are ONE product on gv_made_gradw_f32 (+ its split sum)
        fused_gradw = (ctx.chain and MADE_GRADW_F32 and S > 0 and all(w_ % 4 == 0 for w_ in widths + kin)
                       and all(ctx.needs_input_grad[4 + l] for l in range(L))
                       and all(ctx.needs_input_grad[4 + L + l] or not ctx.has_bias[l] for l in range(L)))
        row_bwd = fused_gradw and ctx.row           # pass 0's backward chain as one single-workgroup launch: gm_l = its masked row gradients
        if row_bwd:     # (launched below, with the products that consume it: nothing on the path to dL/dz reads these rows)
            rows0 = [torch.empty(1, widths[l], **f32) for l in range(L)]
        for l in reversed(range(L)) if not row_bwd else ():
            rows0[l] = g_row
            if l > 0:
                mask = acts0[l] if l < L - 1 else None
                g_row = gemm(g_row, ws[l], a_relu_mask=mask)
        # dL/dW_l, dL/db_l.  Nothing later in the backward pass needs them: with every one of them going straight into the optimiser's
        # arena they run on the side stream, beside the next flow's (launch-bound) passes
        _verify_direct(ctx)
        wants_w = [ctx.needs_input_grad[4 + l] for l in range(L)]
        wants_b = [ctx.has_bias[l] and ctx.needs_input_grad[4 + L + l] for l in range(L)]
        tgt_w = [ctx.direct_w[l] if (wants_w[l] and ctx.direct_w[l] is not None and ctx.direct_w[l].data_ptr() in GRAD_FRESH
                                     and ctx.direct_w[l].is_contiguous()) else None for l in range(L)]
        tgt_b = [ctx.direct_b[l] if (wants_b[l] and ctx.direct_b[l] is not None and ctx.direct_b[l].is_contiguous()) else None
                 for l in range(L)]
        beside = (ctx.masks is not None and L <= 8 and all(tgt_w[l] is not None for l in range(L) if wants_w[l])
                  and all(tgt_b[l] is not None for l in range(L) if wants_b[l]))
        g_ws, g_bs = [], []
        with backward_side(beside, grads, xin, acts, acts0, rows0, zero_row, g_row, ws):
            if row_bwd:
                made_row_bwd(g_row, [dict(w=ws[l], act=acts0[l] if l < L - 1 else None, gb=rows0[l]) for l in range(L)], exact=True)
            if fused_gradw:     # all layers' products in one launch pair
                both = made_gradw_f32_multi([dict(g=grads[l], a=xin if l == 0 else acts[l - 1],
                                                  wmask=ctx.masks[l] if ctx.masks is not None else None, g0=rows0[l],
                                                  g0_act=acts0[l] if (l < L - 1 and not row_bwd) else None,
                                                  a0=zero_row if l == 0 else acts0[l - 1], out=tgt_w[l], accumulate=False, db=tgt_b[l],
                                                  db_accumulate=tgt_b[l] is not None, want_db=wants_b[l]) for l in range(L)])
            for l in range(L) if fused_gradw else ():
                gw, gb = both[l]
                if tgt_w[l] is not None:
                    GRAD_FRESH.discard(tgt_w[l].data_ptr())
                    gw = None
                if tgt_b[l] is not None:
                    GRAD_FRESH.discard(tgt_b[l].data_ptr())
                    gb = None
                g_ws.append(gw)
                g_bs.append(gb)
            for l in range(L) if not fused_gradw else ():
                mask = acts[l] if l < L - 1 else None
                mask0 = acts0[l] if l < L - 1 else None
                inp0 = zero_row if l == 0 else acts0[l - 1]
                gw = gb = None
                if wants_w[l]:
                    gw = gemm(rows0[l], inp0, trans_a=True, a_relu_mask=mask0)           # pass 0: outer product of two rows
                    if S > 0:
                        inp = xin if l == 0 else acts[l - 1]
                        tiles = None if (words is None or ctx.masks is None) else words[l]['tiles']
                        part = gemm(grads[l], inp, trans_a=True, a_relu_mask=mask,
                                    split_k=pick_split_k(ws[l].shape[0], ws[l].shape[1], S * n), c_tiles=tiles)
                        lib.call('gv_axpby', gw.numel(), None, 1.0, ptr(part), 1.0, ptr(gw), lib.stream())
                if wants_b[l]:      # (a bias with a slice of the arena: ADDED there, whatever the slice holds)
                    gb = colsum(rows0[l], relu_mask=mask0, out=tgt_b[l], accumulate=tgt_b[l] is not None)
                    if S > 0:
                        colsum(grads[l], relu_mask=mask, out=gb, accumulate=True)
                    if tgt_b[l] is not None:
                        GRAD_FRESH.discard(gb.data_ptr())
                        gb = None
                g_ws.append(gw)
                g_bs.append(gb)
            if ctx.masks is not None and not fused_gradw:       # dL/dW = mask * dL/d(mask * W): one launch for all layers, straight into the arena where fresh
                idx = [l for l in range(L) if g_ws[l] is not None]
                for i0 in range(0, len(idx), 8):
                    part_idx = idx[i0:i0 + 8]
                    res = mul_multi([ctx.masks[l] for l in part_idx], [g_ws[l] for l in part_idx], outs=[tgt_w[l] for l in part_idx])
                    for l, r in zip(part_idx, res):
                        if tgt_w[l] is not None:
                            GRAD_FRESH.discard(tgt_w[l].data_ptr())
                            g_ws[l] = None
                        else:
                            g_ws[l] = r
        return (g_z, None, None, None, *g_ws, *g_bs)


# ---- K4 in bf16 (csrc/k_made.hip): bf16 storage of weights and activations, every product the same NT kernel ----------------
def _pad8(n):
    return (int(n) + 7) // 8 * 8


def _empty_t_padded(widths, passes, n, npad, kw):
    """Transposed-copy buffers (width_i, passes*npad), carved out of ONE allocation so that one fill zeroes the pad columns
    [n, npad) of every pass of every buffer (they take part in the weight-gradient reduction); the data columns are written
    by the producing kernels."""
    t = torch.empty(sum(widths), passes * npad, **kw)
    if npad > n:
        t.view(sum(widths), passes, npad)[:, :, n:].zero_()
    out, o = [], 0
    for w in widths:
        out.append(t[o:o + w])
        o += w
    out.append(t)          # last: the whole allocation (one row-sum pass over all of it)
    return out


def _empty_t_tiles(widths, passes, n, kw):
    """Transposed-copy buffers in tiles of 64 rows: ONE allocation [passes * T][sum(widths)][64] (T = ceil(n / 64) tiles per
    pass), buffer i = the column range of width_i -- element (column c, stacked row r) at [r // 64][c][r % 64].  A workgroup of the
    weight-gradient product (gv_gemm_bf16_gradw_tiles) then reads contiguous 128-B x width blocks, and a chain workgroup (64
    rows) writes one.  The rows [n, 64 T) of every pass's last tile are zero (they take part in the reduction): written so by the producers.
    Returns (buffers, tiles per pass, elements between two tiles)."""
    T = (int(n) + 63) // 64
    t = torch.empty(passes * T, sum(widths), 64, **kw)
    # (no fill: every producer of a tile -- the chains' epilogues, the IAF update kernels -- writes zeros into the rows past n;
    # GV_MADE_POISON=1 starts the buffer as NaNs so that a producer that does not shows up in the tests)
    if _os.environ.get('GV_MADE_POISON') == '1':
        t.fill_(float('nan'))
    out, o = [], 0
    for w in widths:
        out.append(t[:, o:o + w])
        o += w
    return out, T, sum(widths) * 64


def cast_bf16(x, y=None, y_t=None):
    """y = bf16(x) row-major and / or y_t[c, r] = bf16(x[r, c]); x fp32 (rows, cols) with unit inner stride."""
    x, ldx = _row_major(x, 'x')
    rows, cols = x.shape
    lib.call('gv_cast_bf16', ptr(x), ldx, rows, cols, ptr(y), y.stride(0) if y is not None else 0, ptr(y_t),
             y_t.stride(0) if y_t is not None else 0, lib.stream())


def gemm_bf16_nt(a, b, m, n, k, bias=None, relu=False, mask=None, c_f32=None, accumulate=False, c_bf16=None, c_bf16_t=None,
                 split_k=1):
    """C[m, n] = epilogue(A[m, k] @ B[n, k]^T) on gv_gemm_bf16_nt; A bf16 or fp32 (rounded while staged), B bf16; row strides
    are taken from the tensors (views of stacked / padded buffers)."""
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = int(lib.load().gv_gemm_bf16_nt_workspace_bytes(m, n, split_k))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
    lib.call('gv_gemm_bf16_nt', ptr(a), 1 if a.dtype == torch.float32 else 0, a.stride(0), ptr(b), b.stride(0), m, n, k,
             ptr(bias), 1 if relu else 0, ptr(mask), mask.stride(0) if mask is not None else 0, ptr(c_f32),
             c_f32.stride(0) if c_f32 is not None else 0, 1 if accumulate else 0, ptr(c_bf16),
             c_bf16.stride(0) if c_bf16 is not None else 0, ptr(c_bf16_t), c_bf16_t.stride(0) if c_bf16_t is not None else 0,
             split_k, ptr(ws), ws_bytes, lib.stream())


DENSE_BF16_NT = _os.environ.get('GV_DENSE_BF16_NT', '1') == '1'


def dense_bf16_forms(w):
    """bf16 copies of a dense weight W (k_in, n_out) for the self-loop products on gv_gemm_bf16_nt (one launch): (W^T as (n_out,
    pad8(k_in)), W as (k_in, pad8(n_out))); None when the bf16 dense path does not apply (fp32 precision, small or odd shapes)."""
    if not (DENSE_BF16_NT and _ops.GEMM_PRECISION == 'bf16' and w.dim() == 2 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0):
        return None
    kin, nout = w.shape
    wt = torch.empty(nout, _pad8(kin), dtype=torch.bfloat16, device=w.device)
    wb = torch.empty(kin, _pad8(nout), dtype=torch.bfloat16, device=w.device)
    cast_bf16(w, wb, wt)
    return wt, wb


def dense_bf16(a, b_nt, n, k, bias=None):
    """a (m, k) fp32 @ b_nt (n, k)^T bf16 -> (m, n) fp32: operands rounded to bf16, fp32 accumulation (gv_gemm_bf16_nt; the tiled
    gv_gemm_bf16 takes 63 us for 40 943 x 200 x 200, this kernel 28)."""
    a, _ = _row_major(a, 'a')
    out = torch.empty(a.shape[0], n, dtype=torch.float32, device=a.device)
    gemm_bf16_nt(a, b_nt, a.shape[0], n, k, bias=bias, c_f32=out)
    return out


def gemm_bf16_gradw_fits(m, n, k, split_k):
    return bool(lib.load().gv_gemm_bf16_gradw_fits(int(m), int(n), int(k), int(split_k)))


def gemm_bf16_gradw(a, b, m, n, k, c_f32, accumulate=True, a_rowsum=None, split_k=2):
    """c_f32 (+)= A[m, k] @ B[n, k]^T over a long reduction on the whole-output kernel (gv_gemm_bf16_gradw), and, from the same
    pass over A, a_rowsum[i] += sum_k A[i, k] (the bias gradient next to the weight gradient).  c_f32 dense (m, n)."""
    if c_f32.stride(0) != n:
        raise ValueError('gemm_bf16_gradw: the result must be dense')
    ws_bytes = int(lib.load().gv_gemm_bf16_gradw_workspace_bytes(m, n, split_k))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
    lib.call('gv_gemm_bf16_gradw', ptr(a), a.stride(0), ptr(b), b.stride(0), m, n, k, ptr(c_f32), 1 if accumulate else 0,
             ptr(a_rowsum), split_k, ptr(ws), ws_bytes, lib.stream())


class _ChainLayer(_ct.Structure):
    """gv_chain_layer of include/gcnvae.h."""
    _fields_ = [('w_packed', _ct.c_void_p), ('bias', _ct.c_void_p), ('mask', _ct.c_void_p), ('out_bf16', _ct.c_void_p),
                ('out_bf16_t', _ct.c_void_p), ('out_f32', _ct.c_void_p), ('n', _ct.c_int32), ('k', _ct.c_int32),
                ('relu', _ct.c_int32), ('accumulate', _ct.c_int32), ('ldmask', _ct.c_int32), ('ldb', _ct.c_int32),
                ('ldt', _ct.c_int32), ('ldc', _ct.c_int32), ('iaf_z', _ct.c_void_p), ('iaf_x_old', _ct.c_void_p),
                ('iaf_colcount', _ct.c_void_p), ('iaf_x_new', _ct.c_void_p), ('iaf_ex', _ct.c_void_p), ('iaf_alpha', _ct.c_void_p),
                ('iaf_ld', _ct.c_int32), ('iaf_reserved', _ct.c_int32), ('iaf_keep_colcount', _ct.c_void_p), ('mask_t', _ct.c_void_p),
                ('add_src', _ct.c_void_p), ('add_colcount', _ct.c_void_p), ('ldmask_t', _ct.c_int32), ('ldbits', _ct.c_int32),
                ('out_bits', _ct.c_void_p), ('mask_bits', _ct.c_void_p), ('x_dup_half', _ct.c_int32), ('t_tile', _ct.c_int32)]


class _ChainIafb(_ct.Structure):
    """gv_chain_iafb of include/gcnvae.h."""
    _fields_ = [('z', _ct.c_void_p), ('ex', _ct.c_void_p), ('gx', _ct.c_void_p), ('gld', _ct.c_void_p), ('gz', _ct.c_void_p),
                ('colcount', _ct.c_void_p), ('gnt', _ct.c_void_p), ('ld', _ct.c_int32), ('d', _ct.c_int32), ('t_tile', _ct.c_int32),
                ('flags', _ct.c_int32), ('n_passes', _ct.c_int32), ('rows_step', _ct.c_int32), ('tiles_step', _ct.c_int32),
                ('cc_step', _ct.c_int32), ('of_step', _ct.c_int64)]


class _ChainFwdPass(_ct.Structure):
    """gv_chain_fwd_pass of include/gcnvae.h."""
    _fields_ = [('x_old', _ct.c_void_p), ('x_new', _ct.c_void_p), ('ex', _ct.c_void_p), ('alpha', _ct.c_void_p),
                ('colcount', _ct.c_void_p), ('keep_colcount', _ct.c_void_p), ('out_bf16', _ct.c_void_p), ('out_bf16_t', _ct.c_void_p),
                ('act_t', _ct.c_void_p * 8), ('act_bits', _ct.c_void_p * 8), ('flags', _ct.c_int32), ('reserved', _ct.c_int32)]


class _ChainFwdRow0(_ct.Structure):
    """gv_chain_fwd_row0 of include/gcnvae.h."""
    _fields_ = [('net_row', _ct.c_void_p), ('x_f32', _ct.c_void_p), ('x_t', _ct.c_void_p)]


class _RowLayer(_ct.Structure):
    """gv_row_layer of include/gcnvae.h."""
    _fields_ = [('w', _ct.c_void_p), ('bias', _ct.c_void_p), ('act', _ct.c_void_p), ('inp', _ct.c_void_p), ('out', _ct.c_void_p),
                ('gw', _ct.c_void_p), ('gb', _ct.c_void_p), ('n', _ct.c_int32), ('k', _ct.c_int32), ('ld', _ct.c_int32),
                ('relu', _ct.c_int32), ('ldgw', _ct.c_int32), ('reserved', _ct.c_int32)]


def _row_layers(layers, exact=False):
    arr = (_RowLayer * len(layers))()
    arr[0].reserved = 1 if exact else 0          # exact fp32 operands instead of the bf16 rounding (the fp32 node)
    for c, d in zip(arr, layers):
        w, ld = _row_major(d['w'], 'w')
        gw = d.get('gw')
        c.w, c.ld, c.n, c.k = ptr(w), ld, w.shape[0], w.shape[1]
        c.bias, c.act, c.inp, c.out = ptr(d.get('bias')), ptr(d.get('act')), ptr(d.get('inp')), ptr(d.get('out'))
        c.gw, c.gb = ptr(gw), ptr(d.get('gb'))
        c.relu = 1 if d.get('relu') else 0
        c.ldgw = gw.stride(0) if gw is not None else 0
    return arr


def made_row_fwd(x, layers, exact=False):
    """The masked MLP on ONE row (MADE's pass 0), one single-workgroup launch: layers = dicts with w (n, k) and optional bias,
    relu, out (n,).  exact: fp32 operands (default: rounded to bf16, configs[2])."""
    arr = _row_layers(layers, exact)
    lib.call('gv_made_row_fwd', ptr(x), len(layers), _ct.addressof(arr), lib.stream())


def made_row_bwd(g_out, layers, g_x=None, exact=False):
    """Backward of made_row_fwd: layers = dicts with w and optional act (ReLU mask), inp (the layer's input row), gw, gb."""
    arr = _row_layers(layers, exact)
    lib.call('gv_made_row_bwd', ptr(g_out), len(layers), _ct.addressof(arr), ptr(g_x), lib.stream())


MADE_ROW = _os.environ.get('GV_MADE_ROW', '1') == '1'
MADE_CHAIN = _os.environ.get('GV_MADE_CHAIN', '1') == '1'


def made_pack_weight(w, fwd=True, bwd=True):
    """Both fragment-packed bf16 copies of a fp32 weight W (n, k): for B = W (a forward layer) and B = W^T (backward-x)."""
    w, ld = _row_major(w, 'w')
    n, k = w.shape
    l = lib.load()
    pf = torch.empty(int(l.gv_made_pack_weight_elems(n, k)), dtype=torch.bfloat16, device=w.device) if fwd else None
    pb = torch.empty(int(l.gv_made_pack_weight_elems(k, n)), dtype=torch.bfloat16, device=w.device) if bwd else None
    lib.call('gv_made_pack_weight', ptr(w), ld, n, k, ptr(pf), ptr(pb), lib.stream())
    return pf, pb


def made_pack_weight_iaf(w):
    """Forward packing of a [mu | alpha] layer whose chain carries the IAF update (gv_made_pack_weight_iaf)."""
    w, ld = _row_major(w, 'w')
    n, k = w.shape
    pf = torch.empty(int(lib.load().gv_made_pack_weight_elems(n, k)), dtype=torch.bfloat16, device=w.device)
    lib.call('gv_made_pack_weight_iaf', ptr(w), ld, n, k, ptr(pf), lib.stream())
    return pf


def made_pack_weights(ws, iaf_last=False):
    """made_pack_weight for every layer of a MADE (at most 8) in one launch: [(packed W, packed W^T), ...]; iaf_last: the last
    layer's forward copy in the tile order of a chain that carries the IAF update."""
    ws = [_row_major(w, 'w') for w in ws]
    l = lib.load()
    dev = ws[0][0].device
    pf = [torch.empty(int(l.gv_made_pack_weight_elems(w.shape[0], w.shape[1])), dtype=torch.bfloat16, device=dev) for w, _ in ws]
    pb = [torch.empty(int(l.gv_made_pack_weight_elems(w.shape[1], w.shape[0])), dtype=torch.bfloat16, device=dev) for w, _ in ws]
    k = len(ws)
    tp = lambda ts: (_ct.c_void_p * k)(*[ptr(t) for t in ts])
    ti = lambda vs: (_ct.c_int32 * k)(*[int(v) for v in vs])
    tw, tf, tb = tp([w for w, _ in ws]), tp(pf), tp(pb)
    tl, tn, tk = ti([ld for _, ld in ws]), ti([w.shape[0] for w, _ in ws]), ti([w.shape[1] for w, _ in ws])
    lib.call('gv_made_pack_weight_multi_iaf' if iaf_last else 'gv_made_pack_weight_multi', k, _ct.addressof(tw), _ct.addressof(tl), _ct.addressof(tn), _ct.addressof(tk),
             _ct.addressof(tf), _ct.addressof(tb), lib.stream())
    return list(zip(pf, pb))


def made_chain_fits(widths_n, widths_k, any_mask):
    nl = len(widths_n)
    arr_n = (_ct.c_int32 * nl)(*[int(v) for v in widths_n])
    arr_k = (_ct.c_int32 * nl)(*[int(v) for v in widths_k])
    return bool(lib.load().gv_made_chain_fits(nl, _ct.addressof(arr_n), _ct.addressof(arr_k), 1 if any_mask else 0))


MADE_CHAIN_FLOPS = {}        # tag -> MEAN flops of a launch (filled while a KernelTimer is installed: bench.py's K4 roofline line)
_chain_flops_acc = {}        # tag -> [sum, launches]: the launches of one tag differ in size where the passes go in groups (3 + 2)


MADE_CHAIN_BYTES = {}        # tag -> MEAN algorithmic HBM bytes of a launch (bf16 chains: what a pass must read and write once)


def _note_chain_flops(tag, flops, nbytes=None):
    acc = _chain_flops_acc.setdefault(tag, [0.0, 0, 0.0])
    acc[0] += flops
    acc[1] += 1
    MADE_CHAIN_FLOPS[tag] = acc[0] / acc[1]
    if nbytes is not None:
        acc[2] += nbytes
        MADE_CHAIN_BYTES[tag] = acc[2] / acc[1]


def _chain_layer_bytes(m, d):
    """Algorithmic bytes one layer dict of made_chain moves for m rows: its outputs and masks once (weights, biases and column
    counts stay in cache; an add source / x_old is read for the columns of count 0 alone: not counted)."""
    n = int(d['n'])
    iaf = d.get('iaf')
    wide = n // 2 if iaf is not None else n               # an IAF layer's bf16 outputs hold x_new: d columns
    b = 0
    b += 2 * wide * m if d.get('out_bf16') is not None else 0
    b += 2 * wide * m if d.get('out_bf16_t') is not None else 0
    b += 4 * n * m if d.get('out_f32') is not None else 0
    b += 2 * n * m if (d.get('mask') is not None or d.get('mask_t') is not None) else 0
    b += 4 * ((n + 31) // 32) * m if (d.get('out_bits') is not None or d.get('mask_bits') is not None) else 0
    if iaf is not None:                                    # z read; ex / alpha written (x_new / x_old: handed-through columns alone)
        b += 4 * wide * m * (1 + (iaf.get('ex') is not None) + (iaf.get('alpha') is not None))
    return b


def made_chain(x, m, layers, tag=None, stage=None):
    """One launch for a chain of NT products (gv_made_chain): layers = dicts with w_packed, n, k and optional bias, relu, mask,
    out_bf16, out_bf16_t (t_tile: in tiles of 64 rows, that many elements apart), out_f32, accumulate.  Row strides are taken
    from the tensors.  stage (x = None then): the IAF update's backward as the chain's first stage (gv_made_chain_iafb) --
    dict(z, ex, gx, gz, colcount, gnt, t_tile, gld=None, overwrite_gz=False), fp32 (m, d) operands of one row stride; with
    passes=dict(n, rows_step, tiles_step, cc_step, of_step) the launch walks n passes whose operands lie those steps apart."""
    if tag is not None and lib.TIMER is not None:
        nb = sum(_chain_layer_bytes(int(m), d) for d in layers)
        if stage is not None:      # gx, ex, z read, g_z written (and read unless it starts here), [g_mu | g_alpha]'s tiled copy written
            dd = int(stage['z'].shape[1])
            nb += int(m) * dd * (4 * (4 if stage.get('overwrite_gz') else 5) + 2 * 2) * int((stage.get('passes') or {}).get('n', 1))
            nb += (int((stage.get('passes') or {}).get('n', 1)) - 1) * sum(_chain_layer_bytes(int(m), d) for d in layers)
        else:
            nb += 2 * int(layers[0]['k']) * int(m) // (2 if layers[0].get('x_dup_half') else 1)
        _note_chain_flops(tag, 2.0 * int(m) * sum(int(d['n']) * int(d['k']) for d in layers) * int(((stage or {}).get('passes') or {}).get('n', 1)), nb)
    arr = _chain_layers(layers)
    if stage is not None:
        ts = [stage[k_] for k_ in ('z', 'ex', 'gx', 'gz')]
        if len({t.stride(0) for t in ts}) != 1:
            raise ValueError('made_chain: the stage operands share one row stride')
        sb = _ChainIafb()
        sb.z, sb.ex, sb.gx, sb.gz = (ptr(t) for t in ts)
        sb.gld, sb.colcount, sb.gnt = ptr(stage.get('gld')), ptr(stage['colcount']), ptr(stage['gnt'])
        sb.ld, sb.d, sb.t_tile = ts[0].stride(0), int(ts[0].shape[1]), int(stage['t_tile'])
        sb.flags = (1 if stage.get('overwrite_gz') else 0) | (2 if stage.get('gx_reversed') else 0)
        ps = stage.get('passes')
        if ps is not None:
            sb.n_passes, sb.rows_step, sb.tiles_step = int(ps['n']), int(ps['rows_step']), int(ps['tiles_step'])
            sb.cc_step, sb.of_step = int(ps['cc_step']), int(ps['of_step'])
        lib.call('gv_made_chain_iafb', _ct.addressof(sb), int(m), len(layers), _ct.addressof(arr), lib.stream(), tag=tag)
        return
    lib.call('gv_made_chain', ptr(x), x.stride(0), int(m), len(layers), _ct.addressof(arr), lib.stream(), tag=tag)


def made_chain_fwd(x, m, layers, passes, tag=None, row0=None):
    """ALL passes of a MADE's forward in one launch (gv_made_chain_fwd).  layers: what made_chain takes for ONE pass (the first
    pass's dicts: strides and shared fields are read from them); passes: per pass dict(x_old, colcount, ex, x_new=None, alpha=None,
    keep=None, out_bf16=None, out_bf16_t=None, act_t=[...], act_bits=[...]) -- the pointers that differ from pass to pass.
    row0 (x = None then): dict(net_row, x_f32, x_t) -- pass 0's update as the launch's first stage (gv_made_chain_fwd_row0)."""
    if tag is not None and lib.TIMER is not None:
        dd = int(layers[-1]['n']) // 2
        # x: read for the first pass alone -- or made from z by the row-0 stage (z read, x written as fp32 and as its tiled copy)
        nb = (2 * int(layers[0]['k']) if row0 is None else (4 + 4 + 2) * dd) * int(m)
        for e in passes:
            nb += sum(2 * int(d['n']) * int(m) + 4 * ((int(d['n']) + 31) // 32) * int(m) for d in layers[:-1])      # tiled copies + sign words
            nb += 4 * dd * int(m) * (1 + 1 + (e.get('alpha') is not None))                                         # z read, ex (alpha) written
            nb += 2 * dd * int(m) * ((e.get('out_bf16') is not None) + (e.get('out_bf16_t') is not None))
        _note_chain_flops(tag, 2.0 * int(m) * len(passes) * sum(int(d['n']) * int(d['k']) for d in layers), nb)
    arr = _chain_layers(layers)
    tab = (_ChainFwdPass * len(passes))()
    for c, d in zip(tab, passes):
        c.x_old, c.x_new, c.ex, c.alpha = ptr(d['x_old']), ptr(d.get('x_new')), ptr(d['ex']), ptr(d.get('alpha'))
        c.colcount, c.keep_colcount = ptr(d['colcount']), ptr(d.get('keep'))
        c.out_bf16, c.out_bf16_t = ptr(d.get('out_bf16')), ptr(d.get('out_bf16_t'))
        for l, (t, b) in enumerate(zip(d['act_t'], d['act_bits'])):
            c.act_t[l], c.act_bits[l] = ptr(t), ptr(b)
        c.flags = 1 if d.get('reverse_x_new') else 0
    if row0 is not None:
        f = _ChainFwdRow0()
        f.net_row, f.x_f32, f.x_t = ptr(row0['net_row']), ptr(row0['x_f32']), ptr(row0['x_t'])
        lib.call('gv_made_chain_fwd_row0', _ct.addressof(f), int(m), len(layers), _ct.addressof(arr), len(passes), _ct.addressof(tab),
                 lib.stream(), tag=tag)
        return
    lib.call('gv_made_chain_fwd', ptr(x), x.stride(0), int(m), len(layers), _ct.addressof(arr), len(passes), _ct.addressof(tab), lib.stream(),
             tag=tag)


def _chain_layers(layers):
    """The gv_chain_layer array of made_chain's layer dicts."""
    arr = (_ChainLayer * len(layers))()
    for c, d in zip(arr, layers):
        mask, ob, ot, of = d.get('mask'), d.get('out_bf16'), d.get('out_bf16_t'), d.get('out_f32')
        c.w_packed, c.bias, c.mask = ptr(d['w_packed']), ptr(d.get('bias')), ptr(mask)
        c.out_bf16, c.out_bf16_t, c.out_f32 = ptr(ob), ptr(ot), ptr(of)
        c.n, c.k, c.relu, c.accumulate = int(d['n']), int(d['k']), 1 if d.get('relu') else 0, 1 if d.get('accumulate') else 0
        c.ldmask = mask.stride(0) if mask is not None else 0
        c.ldb = ob.stride(0) if ob is not None else 0
        c.ldt = ot.stride(0) if ot is not None else 0
        if ot is not None and d.get('t_tile'):      # out_bf16_t in tiles of 64 rows (what gemm_bf16_gradw_tiles reads): ot starts at tile 0
            c.ldt, c.t_tile = 64, int(d['t_tile'])
        c.ldc = of.stride(0) if of is not None else 0
        iaf = d.get('iaf')
        if iaf is not None:      # dict(z, x_old, colcount, x_new=None, ex=None, alpha=None): fp32 [m][ld] with one common row stride
            ts = [iaf[k_] for k_ in ('z', 'x_old', 'x_new', 'ex', 'alpha') if iaf.get(k_) is not None]
            if len({t.stride(0) for t in ts}) != 1:
                raise ValueError('made_chain: the IAF operands share one row stride')
            c.iaf_z, c.iaf_x_old, c.iaf_colcount = ptr(iaf['z']), ptr(iaf['x_old']), ptr(iaf['colcount'])
            c.iaf_x_new, c.iaf_ex, c.iaf_alpha = ptr(iaf.get('x_new')), ptr(iaf.get('ex')), ptr(iaf.get('alpha'))
            c.iaf_ld = ts[0].stride(0)
            c.iaf_reserved = int(iaf.get('debug', 0))
            c.iaf_keep_colcount = ptr(iaf.get('keep'))
        mt_, add = d.get('mask_t'), d.get('add')
        if mt_ is not None:
            c.mask_t, c.ldmask_t = ptr(mt_), mt_.stride(0)
        c.x_dup_half = 1 if d.get('x_dup_half') else 0
        obits, mbits = d.get('out_bits'), d.get('mask_bits')       # int32 (m, >= ceil(n / 32)) sign bits of a hidden activation
        if obits is not None or mbits is not None:
            bt = obits if obits is not None else mbits
            if bt.dtype != torch.int32:
                raise ValueError('made_chain: mask bits are int32 words')
            c.out_bits, c.mask_bits, c.ldbits = ptr(obits), ptr(mbits), bt.stride(0)
        if add is not None:      # (src fp32 [m][ldc], colcount): out_f32 += src where colcount == 0
            if of is None or add[0].stride(0) != of.stride(0):
                raise ValueError('made_chain: add_src shares the row stride of out_f32')
            c.add_src, c.add_colcount = ptr(add[0]), ptr(add[1])
    return arr


def gemm_bf16_gradw_tiles(a, a_tile, b, b_tile, m, n, k, c_f32, accumulate=True, a_rowsum=None, split_k=2):
    """gemm_bf16_gradw with both operands in 64-deep K tiles (gv_gemm_bf16_gradw_tiles): element (row, kk) of A at
    a.flatten()[(kk // 64) * a_tile + row * 64 + kk % 64]; a, b: bf16 tensors whose first element is tile 0 of row 0."""
    if c_f32.stride(0) != n:
        raise ValueError('gemm_bf16_gradw_tiles: the result must be dense')
    ws_bytes = int(lib.load().gv_gemm_bf16_gradw_workspace_bytes(m, n, split_k))
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=c_f32.device)
    lib.call('gv_gemm_bf16_gradw_tiles', ptr(a), int(a_tile), ptr(b), int(b_tile), m, n, k, ptr(c_f32), 1 if accumulate else 0,
             ptr(a_rowsum), split_k, ptr(ws), ws_bytes, lib.stream())


# ---- K4 fused in fp32 (csrc/k_chain32.hip): one launch per MADE pass on the fp32 MFMA, zero groups of the masked weights skipped ----
class _Chain32Layer(_ct.Structure):
    """gv_chain32_layer of include/gcnvae.h."""
    _fields_ = [('w_packed', _ct.c_void_p), ('bias', _ct.c_void_p), ('mask', _ct.c_void_p), ('out_f32', _ct.c_void_p),
                ('n', _ct.c_int32), ('k', _ct.c_int32), ('relu', _ct.c_int32), ('accumulate', _ct.c_int32),
                ('ldmask', _ct.c_int32), ('ldc', _ct.c_int32)]


MADE_CHAIN_F32 = _os.environ.get('GV_MADE_CHAIN_F32', '1') == '1'      # the fp32 node's passes as one launch each
MADE_CHAIN_F32_SKIP = _os.environ.get('GV_MADE_CHAIN_F32_SKIP', '1') == '1'      # ... walking only the non-zero groups of the masks
PLAN_WORDS = 4 + 4 * 2 * 8 * 16 + 2 * 8 * 16          # GV_CHAIN32_PLAN_WORDS
_chain32_plans = {}


def made_chain_f32_fits(widths_n, widths_k):
    nl = len(widths_n)
    arr_n = (_ct.c_int32 * nl)(*[int(v) for v in widths_n])
    arr_k = (_ct.c_int32 * nl)(*[int(v) for v in widths_k])
    return bool(lib.load().gv_made_chain_f32_fits(nl, _ct.addressof(arr_n), _ct.addressof(arr_k)))


def made_pack_weights_f32(ws, fwd=True, bwd=True):
    """Fragment-packed fp32 copies of every layer's weight W (n, k) in one launch: [(B = W^T for the forward layer, B = W for the
    backward-x layer), ...] (gv_made_pack_weight_f32_multi)."""
    ws = [_row_major(w, 'w') for w in ws]
    l = lib.load()
    dev = ws[0][0].device
    f32 = dict(dtype=torch.float32, device=dev)
    pf = [torch.empty(int(l.gv_made_pack_weight_f32_elems(w.shape[0], w.shape[1])), **f32) if fwd else None for w, _ in ws]
    pb = [torch.empty(int(l.gv_made_pack_weight_f32_elems(w.shape[1], w.shape[0])), **f32) if bwd else None for w, _ in ws]
    k = len(ws)
    tp = lambda ts: (_ct.c_void_p * k)(*[ptr(t) for t in ts])
    ti = lambda vs: (_ct.c_int32 * k)(*[int(v) for v in vs])
    tw, tf, tb = tp([w for w, _ in ws]), tp(pf), tp(pb)
    tl, tn, tk = ti([ld for _, ld in ws]), ti([w.shape[0] for w, _ in ws]), ti([w.shape[1] for w, _ in ws])
    lib.call('gv_made_pack_weight_f32_multi', k, _ct.addressof(tw), _ct.addressof(tl), _ct.addressof(tn), _ct.addressof(tk),
             _ct.addressof(tf), _ct.addressof(tb), lib.stream())
    return list(zip(pf, pb))


def made_chain_f32_plan(widths_n, widths_k, masks=None, transposed=False):
    """The plan of a chain (gv_made_chain_f32_plan): which 8-deep groups of every 32-column tile hold a non-zero of the layer's 0/1
    mask, and the tiles dealt to the four waves.  One small launch per (mask set, widths, direction), then cached -- the entry
    keeps the mask tensors it was computed from.  masks: per layer the mask of W (n_out, n_in) or None (dense); transposed: the
    layers are backward-x layers (B = W: the layer's k runs over W's rows)."""
    nl = len(widths_n)
    if not MADE_CHAIN_F32_SKIP:
        masks = None
    key = (torch.cuda.current_device(), tuple(int(v) for v in widths_n), tuple(int(v) for v in widths_k), bool(transposed),
           tuple((m.data_ptr(), m._version) if m is not None else None for m in masks) if masks is not None else None)
    hit = _chain32_plans.get(key)
    if hit is not None:
        return hit[0]
    dev = torch.device('cuda', torch.cuda.current_device())
    plan = torch.empty(PLAN_WORDS, dtype=torch.int32, device=dev)
    arr_n = (_ct.c_int32 * nl)(*[int(v) for v in widths_n])
    arr_k = (_ct.c_int32 * nl)(*[int(v) for v in widths_k])
    held, tm, tl = [], None, None
    if masks is not None:
        held = [_row_major(m, 'mask')[0] if m is not None else None for m in masks]
        for m, n_, k_ in zip(held, widths_n, widths_k):
            if m is not None and tuple(m.shape) != ((k_, n_) if transposed else (n_, k_)):
                raise ValueError(f'made_chain_f32_plan: mask {tuple(m.shape)} does not belong to a layer of n={n_}, k={k_}')
        tm = (_ct.c_void_p * nl)(*[ptr(m) for m in held])
        tl = (_ct.c_int32 * nl)(*[m.stride(0) if m is not None else 0 for m in held])
    tt = (_ct.c_int32 * nl)(*[1 if transposed else 0] * nl)
    lib.call('gv_made_chain_f32_plan', nl, _ct.addressof(arr_n), _ct.addressof(arr_k), _ct.addressof(tm) if tm is not None else None,
             _ct.addressof(tl) if tl is not None else None, _ct.addressof(tt), ptr(plan), lib.stream())
    if not _capturing():
        _chain32_plans[key] = (plan, held, masks)
    return plan


def made_chain_f32(x, m, layers, plan, tag=None):
    """One launch for a chain of fp32 products (gv_made_chain_f32): layers = dicts with w_packed, n, k and optional bias, relu,
    mask (fp32, kept where > 0), out_f32, accumulate; row strides are taken from the tensors.  Inside ops.live_rows a chain over
    a node array of exactly ``cap`` rows skips the workgroups that hold only padding rows."""
    if tag is not None and lib.TIMER is not None:
        _note_chain_flops(tag, 2.0 * int(m) * sum(int(d['n']) * int(d['k']) for d in layers))
    arr = (_Chain32Layer * len(layers))()
    for c, d in zip(arr, layers):
        mask, of = d.get('mask'), d.get('out_f32')
        c.w_packed, c.bias, c.mask, c.out_f32 = ptr(d['w_packed']), ptr(d.get('bias')), ptr(mask), ptr(of)
        c.n, c.k, c.relu, c.accumulate = int(d['n']), int(d['k']), 1 if d.get('relu') else 0, 1 if d.get('accumulate') else 0
        c.ldmask = mask.stride(0) if mask is not None else 0
        c.ldc = of.stride(0) if of is not None else 0
    live = _ops.LIVE_ROWS
    rows_dev = live[0] if (live is not None and int(m) == live[1] and x.device == live[0].device) else None
    lib.call('gv_made_chain_f32', ptr(x), x.stride(0), int(m), len(layers), _ct.addressof(arr), ptr(plan), ptr(rows_dev), lib.stream(),
             tag=tag)


MADE_GRADW_F32 = _os.environ.get('GV_MADE_GRADW_F32', '1') == '1'      # the fp32 node's weight / bias gradients on gv_made_gradw_f32
MADE_ROW_F32 = _os.environ.get('GV_MADE_ROW_F32', '1') == '1'          # ... and its pass 0 (one broadcast row) as single-workgroup launches
_gradw32_plans = {}


def made_gradw_f32_plan(m, n, wmask=None):
    """Which 32 x 32 tiles of an (m, n) weight gradient can be non-zero under the layer's 0/1 mask (gv_made_gradw_f32_plan); one
    small launch per mask, cached with the mask it was computed from."""
    key = (torch.cuda.current_device(), int(m), int(n), (wmask.data_ptr(), wmask._version) if wmask is not None else None)
    hit = _gradw32_plans.get(key)
    if hit is not None:
        return hit[0]
    if wmask is not None:
        wmask, ldw = _row_major(wmask, 'wmask')
        if tuple(wmask.shape) != (m, n):
            raise ValueError(f'made_gradw_f32_plan: mask {tuple(wmask.shape)} for an ({m}, {n}) weight')
    plan = torch.empty(int(lib.load().gv_made_gradw_f32_plan_words(m, n)), dtype=torch.int32, device=torch.device('cuda', torch.cuda.current_device()))
    lib.call('gv_made_gradw_f32_plan', ptr(wmask), ldw if wmask is not None else 0, int(m), int(n), ptr(plan), lib.stream())
    if not _capturing():
        _gradw32_plans[key] = (plan, wmask)
    return plan


def made_gradw_f32(g, a, wmask=None, g0=None, g0_act=None, a0=None, out=None, accumulate=False, db=None, db_accumulate=False, want_db=True):
    """(dW, db) of one masked layer over all stacked rows (gv_made_gradw_f32): dW = wmask * (g^T a + g0m^T a0), db = column sums of g
    + g0m, with g0m = pass 0's row gradient g0 (1, m) behind its ReLU mask g0_act (1, m) or None.  g (k, m) and a (k, n) fp32 with
    unit inner stride; ``out`` / ``db``: where to store (or, with accumulate / db_accumulate, add) -- new tensors when None."""
    g, ldg = _row_major(g, 'g')
    a, lda = _row_major(a, 'a')
    k, m = g.shape
    n = a.shape[1]
    if a.shape[0] != k:
        raise ValueError('made_gradw_f32: g and a have different row counts')
    f32 = dict(dtype=torch.float32, device=g.device)
    if out is None:
        out, accumulate = torch.empty(m, n, **f32), False
    if db is None and want_db:
        db, db_accumulate = torch.empty(m, **f32), False
    if out.stride(1) != 1 or (db is not None and not db.is_contiguous()):
        raise ValueError('made_gradw_f32: outputs with unit inner stride')
    plan = made_gradw_f32_plan(m, n, wmask)
    ldw = 0
    if wmask is not None:
        wmask, ldw = _row_major(wmask, 'wmask')
    nws = int(lib.load().gv_made_gradw_f32_workspace_floats(m, n, k))
    ws = torch.empty(nws, **f32)
    lib.call('gv_made_gradw_f32', ptr(g), ldg, ptr(a), lda, int(m), int(n), int(k), ptr(plan), ptr(wmask), ldw, ptr(g0), ptr(g0_act), ptr(a0),
             ptr(out), out.stride(0), 1 if accumulate else 0, ptr(db), 1 if db_accumulate else 0, ptr(ws), nws, lib.stream())
    return out, db


class _GradW32Item(_ct.Structure):
    """gv_gradw32_item of include/gcnvae.h."""
    _fields_ = [(f, _ct.c_void_p) for f in ('g', 'a', 'plan', 'wmask', 'g0', 'g0_act', 'a0', 'out', 'db')] + \
               [(f, _ct.c_int32) for f in ('ldg', 'lda', 'm', 'n', 'ldw', 'ldo', 'accumulate', 'db_accumulate')] + [('k', _ct.c_int64)]


def made_gradw_f32_multi(items):
    """made_gradw_f32 for up to 8 layers in ONE launch pair (gv_made_gradw_f32_multi): ``items`` = dicts with made_gradw_f32's
    arguments; returns [(dW, db), ...]."""
    if not 1 <= len(items) <= 8:
        raise ValueError('made_gradw_f32_multi: 1 .. 8 products')
    arr, keep, res = (_GradW32Item * len(items))(), [], []
    for c, d_ in zip(arr, items):
        g, ldg = _row_major(d_['g'], 'g')
        a, lda = _row_major(d_['a'], 'a')
        k, m = g.shape
        n = a.shape[1]
        if a.shape[0] != k:
            raise ValueError('made_gradw_f32: g and a have different row counts')
        f32 = dict(dtype=torch.float32, device=g.device)
        out, accumulate = d_.get('out'), bool(d_.get('accumulate'))
        db, db_accumulate = d_.get('db'), bool(d_.get('db_accumulate'))
        if out is None:
            out, accumulate = torch.empty(m, n, **f32), False
        if db is None and d_.get('want_db', True):
            db, db_accumulate = torch.empty(m, **f32), False
        if out.stride(1) != 1 or (db is not None and not db.is_contiguous()):
            raise ValueError('made_gradw_f32: outputs with unit inner stride')
        wmask, ldw = d_.get('wmask'), 0
        plan = made_gradw_f32_plan(m, n, wmask)
        if wmask is not None:
            wmask, ldw = _row_major(wmask, 'wmask')
        c.g, c.a, c.plan, c.wmask, c.out, c.db = ptr(g), ptr(a), ptr(plan), ptr(wmask), ptr(out), ptr(db)
        c.g0, c.g0_act, c.a0 = ptr(d_.get('g0')), ptr(d_.get('g0_act')), ptr(d_.get('a0'))
        c.ldg, c.lda, c.m, c.n, c.ldw, c.ldo, c.k = ldg, lda, int(m), int(n), ldw, out.stride(0), int(k)
        c.accumulate, c.db_accumulate = 1 if accumulate else 0, 1 if db_accumulate else 0
        keep.append((g, a, wmask))
        res.append((out, db))
    nws = int(lib.load().gv_made_gradw_f32_multi_workspace_floats(len(items), _ct.addressof(arr)))
    ws = torch.empty(nws, dtype=torch.float32, device=res[0][0].device)
    lib.call('gv_made_gradw_f32_multi', len(items), _ct.addressof(arr), ptr(ws), nws, lib.stream())
    return res


class _Chain32Iaf(_ct.Structure):
    """gv_chain32_iaf of include/gcnvae.h."""
    _fields_ = [('mode', _ct.c_int32), ('passes', _ct.c_int32), ('d', _ct.c_int32), ('flags', _ct.c_int32), ('step', _ct.c_int64),
                ('z', _ct.c_void_p), ('colcount', _ct.c_void_p), ('x_out', _ct.c_void_p), ('net', _ct.c_void_p), ('g_in', _ct.c_void_p),
                ('g_logdet', _ct.c_void_p), ('g_z', _ct.c_void_p), ('ld_net', _ct.c_int32), ('reserved', _ct.c_int32),
                ('net0', _ct.c_void_p), ('cnt0', _ct.c_void_p), ('log_det', _ct.c_void_p)]


MADE_PASSES_F32 = _os.environ.get('GV_MADE_PASSES_F32', '1') == '1'      # all stacked passes of a MADE + the IAF updates in ONE launch per direction


def made_passes_f32(x, m, layers, plan, iaf, tag=None):
    """Several passes of a MADE in one launch with the IAF update fused in (gv_made_passes_f32): ``layers`` as made_chain_f32, for the
    launch's FIRST pass; ``iaf``: dict(mode=1 forward / 2 backward, passes, step (rows between passes in the stacked buffers, signed),
    d, z, colcount (the first pass's counts), flags, and x_out (forward) / net, g_in, g_logdet, g_z (backward))."""
    if tag is not None and lib.TIMER is not None:
        _note_chain_flops(tag, 2.0 * int(m) * int(iaf['passes']) * sum(int(d_['n']) * int(d_['k']) for d_ in layers))
    arr = (_Chain32Layer * len(layers))()
    for c, d_ in zip(arr, layers):
        mask, of = d_.get('mask'), d_.get('out_f32')
        c.w_packed, c.bias, c.mask, c.out_f32 = ptr(d_['w_packed']), ptr(d_.get('bias')), ptr(mask), ptr(of)
        c.n, c.k, c.relu, c.accumulate = int(d_['n']), int(d_['k']), 1 if d_.get('relu') else 0, 1 if d_.get('accumulate') else 0
        c.ldmask = mask.stride(0) if mask is not None else 0
        c.ldc = of.stride(0) if of is not None else 0
    ia = _Chain32Iaf()
    ia.mode, ia.passes, ia.d, ia.flags, ia.step = int(iaf['mode']), int(iaf['passes']), int(iaf['d']), int(iaf.get('flags', 0)), int(iaf['step'])
    ia.z, ia.colcount, ia.x_out = ptr(iaf['z']), ptr(iaf['colcount']), ptr(iaf.get('x_out'))
    net = iaf.get('net')
    ia.net, ia.ld_net = ptr(net), net.stride(0) if net is not None else 0
    ia.g_in, ia.g_logdet, ia.g_z = ptr(iaf.get('g_in')), ptr(iaf.get('g_logdet')), ptr(iaf.get('g_z'))
    ia.net0, ia.cnt0, ia.log_det = ptr(iaf.get('net0')), ptr(iaf.get('cnt0')), ptr(iaf.get('log_det'))
    live = _ops.LIVE_ROWS
    rows_dev = live[0] if (live is not None and int(m) == live[1] and x.device == live[0].device) else None
    lib.call('gv_made_passes_f32', ptr(x), x.stride(0), int(m), len(layers), _ct.addressof(arr), ptr(plan), ptr(rows_dev), _ct.addressof(ia),
             lib.stream(), tag=tag)


def _made_params_work(masks, ws, bs, d, S):
    """The part of a bf16 MADE forward that depends on the parameters alone: the mask fold (one launch for all layers), the
    weights as bf16 -- fragment-packed for the chain kernel, forward and transposed form -- and pass 0 on its single zero row."""
    if masks is not None:
        ws = mul_multi(masks, ws)
    L = len(ws)
    dev = ws[0].device
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    widths = [w.shape[0] for w in ws]
    chain = (MADE_CHAIN and S > 0 and L <= 8 and made_chain_fits(widths, [w.shape[1] for w in ws], False)
             and made_chain_fits([w.shape[1] for w in reversed(ws)], [w.shape[0] for w in reversed(ws)], True))
    fused = chain and MADE_CHAIN_IAF and d % 8 == 0 and widths[L - 1] == 2 * d
    if chain:       # one launch per pass: fragment-packed weights (forward and transposed form from one launch per layer)
        packed = made_pack_weights(ws, iaf_last=fused)
        wbf, wbt = [pk[0] for pk in packed], [pk[1] for pk in packed]
    else:
        wbf = [torch.empty(w.shape[0], _pad8(w.shape[1]), **bf) for w in ws]
        wbt = [torch.empty(w.shape[1], _pad8(w.shape[0]), **bf) for w in ws]
        for w, a_, t_ in zip(ws, wbf, wbt):
            cast_bf16(w, a_, t_)
    # pass 0 on a single zero row (tiny: the generic GEMM with bf16-rounded operands)
    zero_row = _zero_row(d, dev)
    row = MADE_ROW and L <= 8 and d <= 512 and max(widths) <= 512 and all(w.shape[1] % 4 == 0 for w in ws)
    if row:         # one single-workgroup launch for the whole row
        acts0 = [torch.empty(1, widths[l], **f32) for l in range(L)]
        made_row_fwd(None, [dict(w=ws[l], bias=bs[l], relu=l < L - 1, out=acts0[l]) for l in range(L)])
    else:
        acts0, inp = [], zero_row
        for l in range(L):
            inp = gemm(inp, ws[l], trans_b=True, bias=bs[l], act=ACT_RELU if l < L - 1 else ACT_NONE, precision='bf16')
            acts0.append(inp)
    return dict(ws=ws, chain=chain, fused=fused, wbf=wbf, wbt=wbt, row=row, acts0=acts0, zero_row=zero_row)


# A model that knows its MADE calls ahead of time (the IAF stack behind the R-GCN encoder) announces them at the start of its
# forward: made_prepare runs _made_params_work of every announced call on ONE side stream -- beside the encoder's layers; under
# capture a parallel branch -- and the node picks the result up behind an event.  Per flow that is a mask fold, a packing launch,
# two fills and the single-workgroup row kernel: ~70 us that used to sit in front of every flow's first pass.
MADE_PREPARE = _os.environ.get('GV_MADE_PREPARE', '1') == '1'
_made_prep = {}


def _made_prep_key(ws, d, S):
    return (tuple(w.data_ptr() for w in ws), int(d), int(S))


def made_prepare(calls):
    """calls: [(colcount, weights, biases, masks)] of the MADE nodes the caller is about to run (made_forward's arguments)."""
    if not calls or not calls[0][1][0].is_cuda or not _ops.launch_layout().made_prepare:      # (host tensors: the node itself refuses them)
        return
    bf16 = _ops.GEMM_PRECISION == 'bf16' and MADE_BF16_STORAGE
    main, side = torch.cuda.current_stream(), _side('made_prep')
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        done, joined, mine = torch.cuda.Event(), [False], []
        for colcount, weights, biases, masks in calls:
            d, S = weights[0].shape[1], colcount.shape[0] - 1
            if masks is None or len(weights) > 8:
                continue        # (not the call made_forward hands to a node with these very tensors)
            if bf16 and not (d % 8 or any(w.shape[0] % 8 or w.shape[1] % 8 for w in weights)):
                prep = _made_params_work(tuple(masks), tuple(weights), tuple(biases), d, S)
            else:               # the fp32 node
                prep = _made_params_work_f32(tuple(masks), tuple(weights), tuple(biases), d, S)
            prep['done'], prep['joined'] = done, joined
            _made_prep[_made_prep_key(weights, d, S)] = prep
            mine.append(prep)
        if mine:
            done.record(side)         # one event behind all of it: the first node that picks its part up waits, the others need not


def made_prepare_finish():
    """Drop what was prepared and not picked up (and join the side stream: nothing may stay unjoined in a captured step)."""
    if _made_prep:
        torch.cuda.current_stream().wait_stream(_side('made_prep'))
        _made_prep.clear()


class _MADEForwardBF16(torch.autograd.Function):
    """MADE.forward (kgvae/flow_network.py:85-98) with bf16 operands in MEMORY (BASELINE configs[2]; semantics as the
    tests' CPU emulation pins them: operands rounded to bf16, fp32 products and sums).  Same structure as _MADEForward -- pass 0 on one
    broadcast row, the later passes stacked -- but the stacked activations are stored as bf16 row-major PLUS a bf16
    transposed copy (written by the producing GEMM's epilogue), the backward keeps the ReLU-masked gradients of every layer the
    same way, and all three product kinds (forward, backward-x, backward-W) run on gv_gemm_bf16_nt.  Per-pass column offset
    in the transposed buffers is rounded up to 8 rows (8-B aligned stores); the pad columns stay zero."""

    @staticmethod
    def forward(ctx, z, colcount, masks, *wb):
        ctx.set_materialize_grads(False)
        reverse_out, wb = bool(wb[-1]), wb[:-1]      # x comes out with its columns reversed (the PermuteLayer behind the block rides along)
        L = len(wb) // 2
        ws, bs = wb[:L], wb[L:]
        ctx.masks = masks
        if masks is not None:          # raw weights + their masks: folded here, all layers in one launch (and in backward likewise)
            ctx.direct_w = [_direct(w) for w in ws]
        z = _chk(z.contiguous(), name='z')
        n, d = z.shape
        P = colcount.shape[0]
        S = P - 1
        dev = z.device
        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        st = lib.stream()
        npad = _pad8(n)
        # what depends on the parameters alone (mask fold, bf16 / fragment-packed weights, pass 0's row): done ahead on a side stream
        # where the model announced this call (made_prepare), here otherwise
        prep = _made_prep.pop(_made_prep_key(ws, d, S), None)
        prep_was_ready = prep is not None and prep['joined'][0]
        if prep is not None:
            if not prep['joined'][0]:       # ONE join for everything that was prepared (a cross-stream edge costs a replayed graph 20-40 us)
                fork_sync()
                torch.cuda.current_stream().wait_event(prep['done'])
                prep['joined'][0] = True
        else:
            fork_sync()
            prep = _made_params_work(masks, ws, bs, d, S)
        ws, chain, fused, wbf, wbt, row, acts0, zero_row = (prep[k_] for k_ in ('ws', 'chain', 'fused', 'wbf', 'wbt', 'row', 'acts0', 'zero_row'))
        widths = [w.shape[0] for w in ws]                       # layer output widths; inputs: d, then widths[:-1]
        xin = torch.empty(max(S, 1) * n, d, **f32)               # fp32 pass inputs (update pass-through, backward)
        xin_b = torch.empty(max(S, 1) * n, _pad8(d), **bf)
        # fused: the transposed copies (read by the weight-gradient products alone) in tiles of 64 rows, when every product fits
        # the whole-output kernel; pass p then starts at tile p * T
        T = (n + 63) // 64
        split_t = max(2, min(GRADW_SPLIT_MAX, S * T * 64 // 512))
        tiled = (fused and MADE_T_TILES and S > 0 and d % 4 == 0
                 and all(gemm_bf16_gradw_fits(widths[l], ws[l].shape[1], S * T * 64, split_t) for l in range(L)))
        if tiled:
            tbufs, T, tt = _empty_t_tiles([d] + widths[:L - 1], S, n, bf)
        else:
            tbufs, tt = _empty_t_padded([d] + widths[:L - 1], max(S, 1), n, npad, bf)[:-1], 0
        t_of = (lambda buf, q, r0=0: dict(out_bf16_t=buf[q * T + r0 // 64:], t_tile=tt)) if tiled else \
               (lambda buf, q, r0=0: dict(out_bf16_t=buf[:, q * npad:q * npad + n]))
        xin_t = tbufs[0]
        # fused (the IAF update inside the chain): row-major activations never leave the chain (the backward chain stages its ReLU
        # masks from the transposed copies), and of [mu | alpha] only exp(alpha + mu) is kept (+ alpha of the last pass)
        acts_b = [] if fused else [torch.empty(max(S, 1) * n, _pad8(widths[l]), **bf) for l in range(L - 1)]
        # ... as sign BITS, one int32 word per (row, 32 columns)
        sign = [torch.empty(max(S, 1) * n, (widths[l] + 31) // 32, dtype=torch.int32, device=dev) for l in range(L - 1)] if fused else []
        acts_t = tbufs[1:]
        if fused:
            net_out = torch.empty(max(S, 1) * n, d, **f32)           # exp(alpha + mu) of every stacked pass
            alpha_last = torch.empty(n, d, **f32)
        else:
            net_out = torch.empty(max(S, 1) * n, widths[L - 1], **f32)   # [mu | alpha] of every stacked pass
        x_out = torch.empty(n, d, **f32)
        def update(net, ld_net, x_old, cc, q):
            """The IAF update of one pass.  Its result is pass q + 1's input (slice q of the stacked buffers: fp32 + the bf16
            row-major and transposed copies the products read, written by the same launch) or, after the last pass, x_out."""
            if q < S and tiled:
                lib.call('gv_iaf_update_fwd_bf16_tiles', ptr(z), ptr(net), ld_net, ptr(x_old), ptr(cc), ptr(xin[q * n:(q + 1) * n]),
                         ptr(xin_b[q * n:(q + 1) * n]), xin_b.stride(0), ptr(xin_t[q * T:]), tt, n, d, st)
            elif q < S:
                lib.call('gv_iaf_update_fwd_bf16', ptr(z), ptr(net), ld_net, ptr(x_old), ptr(cc), ptr(xin[q * n:(q + 1) * n]),
                         ptr(xin_b[q * n:(q + 1) * n]), xin_b.stride(0), ptr(xin_t[:, q * npad:q * npad + n]), xin_t.stride(0),
                         n, d, st)
            else:
                lib.call('gv_iaf_update_fwd', ptr(z), ptr(net), ld_net, ptr(x_old), ptr(cc), ptr(x_out), n, d, st)
        # pass 0's update: the first stage of the forward passes launch where that launch runs (gv_made_chain_fwd_row0), else its own launch
        per_launch = MADE_FWD_PASSES or (6 if len(_made_row_blocks(n)) == 1 else 3)
        fwd_loop = fused and tiled and L > 1 and d % 8 == 0 and per_launch > 1 and S > 0
        row0_in_launch = fwd_loop and MADE_FWD_ROW0 and n * xin.stride(0) * 4 < (1 << 32) and acts0[L - 1].is_contiguous()
        # the row blocks stay forked over the whole flow stack where nothing of this node runs over all rows but the log-det row sums
        kept = _FORK is not None and _FORK['n'] == n and row0_in_launch and prep_was_ready
        if not kept:
            fork_sync()
        if not row0_in_launch:
            update(acts0[L - 1], 0, z, colcount[0], 0)

        folded = [False]           # the last pass stored x_out reversed already

        def fused_passes(r0, r1):
            """Passes 1 .. P-1 for the rows [r0, r1) (r0 a multiple of 64): every launch of a pass is row-local."""
            todo = list(range(1, P))
            while todo:
                p = todo.pop(0)
                a, b, na, nb_ = (p - 1) * n + r0, (p - 1) * n + r1, p * n + r0, p * n + r1
                # the pass's IAF update in the last layer's epilogue: x_new and its operand copies leave the chain
                head = dict(w_packed=wbf[L - 1], n=widths[L - 1], k=ws[L - 1].shape[1], bias=bs[L - 1],
                            iaf=dict(z=z[r0:r1], x_old=xin[a:b], colcount=colcount[p], ex=net_out[a:b]))
                if fwd_loop:
                    # ... for up to six passes in ONE launch (gv_made_chain_fwd): a workgroup keeps its 64 rows, x_new stays in LDS
                    # as the next pass's input; its row-major bf16 copy goes to memory only behind the last pass of a launch
                    grp = [p] + todo[:per_launch - 1]
                    del todo[:len(grp) - 1]
                    tab = []
                    for g in grp:
                        ga, gb, gna, gnb = (g - 1) * n + r0, (g - 1) * n + r1, g * n + r0, g * n + r1
                        e = dict(x_old=xin[ga:gb], colcount=colcount[g], ex=net_out[ga:gb],
                                 act_t=[acts_t[l][(g - 1) * T + r0 // 64:] for l in range(L - 1)],
                                 act_bits=[sign[l][ga:gb] for l in range(L - 1)])
                        if g < S:
                            e.update(x_new=xin[gna:gnb], keep=colcount[g + 1], out_bf16_t=xin_t[g * T + r0 // 64:],
                                     out_bf16=xin_b[gna:gnb] if g == grp[-1] else None)
                        else:
                            e.update(x_new=x_out[r0:r1], alpha=alpha_last[r0:r1], reverse_x_new=reverse_out)
                            folded[0] = reverse_out
                        tab.append(e)
                    head.update(out_bf16=xin_b[a:b], **t_of(xin_t, p, r0))          # (strides / tile size of x_new's copies)
                    first = dict(net_row=acts0[L - 1], x_f32=xin[r0:r1], x_t=xin_t[r0 // 64:]) if (row0_in_launch and p == 1) else None
                    made_chain_fwd(xin_b[a:b] if first is None else None, r1 - r0,
                                   [dict(w_packed=wbf[l], n=widths[l], k=ws[l].shape[1], bias=bs[l], relu=True, out_bits=sign[l][a:b],
                                         **t_of(acts_t[l], p - 1, r0)) for l in range(L - 1)] + [head], tab, tag='madechain_fwd', row0=first)
                    continue
                if p < S:   # fp32 x_new only where the next pass hands a column through; its operands in bf16
                    head['iaf'].update(x_new=xin[na:nb_], keep=colcount[p + 1])
                    head.update(out_bf16=xin_b[na:nb_], **t_of(xin_t, p, r0))
                else:
                    head['iaf'].update(x_new=x_out[r0:r1], alpha=alpha_last[r0:r1])
                made_chain(xin_b[a:b], r1 - r0, [dict(w_packed=wbf[l], n=widths[l], k=ws[l].shape[1], bias=bs[l], relu=True,
                                                      out_bits=sign[l][a:b], **t_of(acts_t[l], p - 1, r0)) for l in range(L - 1)] + [head],
                           tag='madechain_fwd')
        if fused:
            _by_row_blocks(fused_passes, n, None if tiled else 1, keep_forked=kept)
        for p in range(1, P) if not fused else ():
            sl = slice((p - 1) * n, p * n)
            tsl = slice((p - 1) * npad, (p - 1) * npad + n)
            inp = xin_b[sl]
            if chain:
                head = dict(w_packed=wbf[L - 1], n=widths[L - 1], k=ws[L - 1].shape[1], bias=bs[L - 1])
                head['out_f32'] = net_out[sl]
                made_chain(inp, n, [dict(w_packed=wbf[l], n=widths[l], k=ws[l].shape[1], bias=bs[l], relu=True,
                                         out_bf16=acts_b[l][sl], out_bf16_t=acts_t[l][:, tsl]) for l in range(L - 1)] + [head],
                           tag='madechain_fwd')
            else:
                for l in range(L - 1):
                    gemm_bf16_nt(inp, wbf[l], n, widths[l], ws[l].shape[1], bias=bs[l], relu=True, c_bf16=acts_b[l][sl],
                                 c_bf16_t=acts_t[l][:, tsl])
                    inp = acts_b[l][sl]
                gemm_bf16_nt(inp, wbf[L - 1], n, widths[L - 1], ws[L - 1].shape[1], bias=bs[L - 1], c_f32=net_out[sl])
            update(net_out[sl], 2 * d, xin[sl], colcount[p], p)
        log_det = torch.empty(n, **f32)
        if kept:        # (over all rows: behind the stack's one join)
            _FORK['deferred'].append(lambda: lib.call('gv_rowsum', ptr(alpha_last), d, 0, d, ptr(log_det), n, lib.stream()))
        elif P > 1 and fused:
            lib.call('gv_rowsum', ptr(alpha_last), d, 0, d, ptr(log_det), n, st)
        elif P > 1:
            lib.call('gv_rowsum', ptr(net_out[(S - 1) * n:]), 2 * d, d, d, ptr(log_det), n, st)
        else:
            log_det = acts0[L - 1][:, d:].sum(dim=1).expand(n).contiguous()
        if reverse_out and not folded[0]:
            fork_sync()
            rev = torch.empty_like(x_out)
            lib.call('gv_reverse_cols', ptr(x_out), ptr(rev), n, d, st)
            x_out = rev
        if kept:
            _FORK['hold'].append(dict(locals()))       # nothing of this node is handed back to the allocator before the join
        ctx.reverse_out = reverse_out
        ctx.save_for_backward(z, colcount, xin_t, zero_row, net_out, *(sign if fused else acts_b), *acts_t, *acts0, *wbt, *ws)
        ctx.L = L
        ctx.chain = chain
        ctx.fused = fused
        ctx.tiled, ctx.t_tile = tiled, tt
        ctx.row = row
        ctx.direct_b = [_direct(b) if b is not None else None for b in bs]
        _stamp_direct(ctx)
        ctx.has_bias = [b is not None for b in bs]
        return x_out, log_det

    @staticmethod
    def backward(ctx, gx, gld):
        L = ctx.L
        saved = ctx.saved_tensors
        z, colcount, xin_t, zero_row, net_out = saved[:5]
        o = 5
        acts_b, acts_t = saved[o:o + L - 1], saved[o + L - 1:o + 2 * (L - 1)]      # fused: acts_b holds the sign-bit words instead
        o += 2 * (L - 1)
        acts0, wbt, ws = saved[o:o + L], saved[o + L:o + 2 * L], saved[o + 2 * L:o + 3 * L]
        n, d = z.shape
        P = colcount.shape[0]
        S = P - 1
        dev = z.device
        f32 = dict(dtype=torch.float32, device=dev)
        bf = dict(dtype=torch.bfloat16, device=dev)
        st = lib.stream()
        npad = _pad8(n)
        widths = [w.shape[0] for w in ws]
        gx = torch.zeros(n, d, **f32) if gx is None else _chk(gx.contiguous(), name='gx')
        gld = None if gld is None else _chk(gld.contiguous(), name='gld')
        # dL/dx arrives with reversed columns where the forward folded the PermuteLayer in: the first backward launch reads it that
        # way when it is the chain with the update's backward as its first stage; otherwise it is reversed here
        gx_rev = ctx.reverse_out and (ctx.fused and P > 1 and MADE_CHAIN_IAFB and ctx.tiled and L > 1 and d % 4 == 0
                                      and n * d * 4 < (1 << 32))
        if ctx.reverse_out and not gx_rev:
            rev = torch.empty_like(gx)
            lib.call('gv_reverse_cols', ptr(gx), ptr(rev), n, d, st)
            gx = rev
        # ReLU-masked gradients w.r.t. every layer's pre-activation: bf16 row-major (operand of backward-x) and transposed
        # (operand of backward-W and of the bias sums)
        if ctx.fused:   # only the chain's input [g_mu | g_alpha], one pass at a time; the hidden layers' stay inside the chain
            gm_in = torch.empty(n, _pad8(widths[L - 1]), **bf)
        else:
            gm_b = [torch.empty(max(S, 1) * n, _pad8(widths[l]), **bf) for l in range(L)]
        tiled, T = ctx.tiled, (n + 63) // 64
        if tiled:
            gm_t, _, tb = _empty_t_tiles(widths, S, n, bf)
            gm_t_all = None
            t_of = lambda buf, q, r0=0: dict(out_bf16_t=buf[q * T + r0 // 64:], t_tile=tb)
        else:
            *gm_t, gm_t_all = _empty_t_padded(widths, max(S, 1), n, npad, bf)
            t_of = lambda buf, q, r0=0: dict(out_bf16_t=buf[:, q * npad:q * npad + n])
        g_z = torch.empty(n, d, **f32) if (ctx.fused and P > 1) else torch.zeros(n, d, **f32)      # fused: the first pass writes it
        gz_p = torch.empty(n, d, **f32)
        g_cur = gx
        # dL/dx_old of every pass, stacked (allocated before any fork)
        gold_stack = torch.empty(max(P - 1, 1) * n, d, **f32) if ctx.fused else None
        g_olds = {p: gold_stack[(p - 1) * n:p * n] for p in range(1, P)} if ctx.fused else None

        def fused_passes(r0, r1):
            """The backward of passes P-1 .. 1 for the rows [r0, r1) (r0 a multiple of 64): every launch of a pass is row-local."""
            g_in = gx
            todo = list(reversed(range(1, P)))
            while todo:
                p = todo.pop(0)
                a, b, t0 = (p - 1) * n + r0, (p - 1) * n + r1, r0 // 64
                # from exp(alpha + mu); the gradient handed through to x_old (columns of count 0) is added by the chain's last layer
                # without a log-det gradient (every pass but the last) g_alpha == g_mu: the chain's row-major input holds the g_mu half
                # alone and the chain stages it twice (x_dup_half)
                half = (gld is None or p != P - 1) and widths[L - 1] % 16 == 0 and L > 1
                if MADE_CHAIN_IAFB and tiled and L > 1 and d % 4 == 0 and (r1 - r0) * d * 4 < (1 << 32):
                    # ... as the backward chain's FIRST STAGE: [g_mu | g_alpha] goes straight into layer 0's LDS tile; up to six passes
                    # (this one and the ones below it: their operands lie one pass apart in the stacked buffers) per launch
                    more = min(len(todo), MADE_CHAIN_PASSES - 1) if g_in.stride(0) == gold_stack.stride(0) else 0
                    passes = dict(n=more + 1, rows_step=-n, tiles_step=-T, cc_step=-colcount.stride(0), of_step=-n * gold_stack.stride(0))
                    made_chain(None, r1 - r0,
                               [dict(w_packed=wbt[L - 1], n=widths[L - 2], k=widths[L - 1], mask_bits=acts_b[L - 2][a:b],
                                     **t_of(gm_t[L - 2], p - 1, r0))] +
                               [dict(w_packed=wbt[l], n=widths[l - 1], k=widths[l], mask_bits=acts_b[l - 1][a:b],
                                     **t_of(gm_t[l - 1], p - 1, r0)) for l in reversed(range(1, L - 1))] +
                               [dict(w_packed=wbt[0], n=d, k=widths[0], out_f32=g_olds[p][r0:r1], add=(g_in[r0:r1], colcount[p]))],
                               tag='madechain_bwd',
                               stage=dict(z=z[r0:r1], ex=net_out[a:b], gx=g_in[r0:r1], gz=g_z[r0:r1], colcount=colcount[p],
                                          gnt=gm_t[L - 1][(p - 1) * T + t0:], t_tile=tb,
                                          gld=gld[r0:r1] if (p == P - 1 and gld is not None) else None, overwrite_gz=p == P - 1,
                                          gx_reversed=gx_rev and p == P - 1, passes=passes if more else None))
                    del todo[:more]
                    g_in = g_olds[p - more]
                    continue
                lib.call('gv_iaf_update_bwd_bf16_ex', ptr(z[r0:r1]), ptr(net_out[a:b]), d, ptr(colcount[p]), ptr(g_in[r0:r1]),
                         ptr(gld[r0:r1]) if (p == P - 1 and gld is not None) else None, ptr(g_z[r0:r1]), ptr(gm_in[r0:r1]), gm_in.stride(0),
                         ptr(gm_t[L - 1][(p - 1) * T + t0:] if tiled else gm_t[L - 1][:, (p - 1) * npad:]),
                         tb if tiled else gm_t[L - 1].stride(0), None,
                         (1 if p == P - 1 else 0) | (2 if half else 0) | (4 if tiled else 0), r1 - r0, d, lib.stream())
                first = dict(w_packed=wbt[L - 1], n=widths[L - 2], k=widths[L - 1], mask_bits=acts_b[L - 2][a:b],
                             x_dup_half=half, **t_of(gm_t[L - 2], p - 1, r0)) if L > 1 else None
                made_chain(gm_in[r0:r1], r1 - r0,
                           ([first] if first is not None else []) +
                           [dict(w_packed=wbt[l], n=widths[l - 1], k=widths[l], mask_bits=acts_b[l - 1][a:b],
                                 **t_of(gm_t[l - 1], p - 1, r0)) for l in reversed(range(1, L - 1))] +
                           [dict(w_packed=wbt[0], n=d, k=widths[0], out_f32=g_olds[p][r0:r1], add=(g_in[r0:r1], colcount[p]))],
                           tag='madechain_bwd')
                g_in = g_olds[p]
        if ctx.fused:
            _by_row_blocks(fused_passes, n, None if tiled else 1)
            if P > 1:
                g_cur = g_olds[1]
        for p in reversed(range(1, P)) if not ctx.fused else ():
            sl = slice((p - 1) * n, p * n)
            tsl = slice((p - 1) * npad, (p - 1) * npad + n)
            g_old = torch.empty(n, d, **f32)
            # the update's backward: g_z accumulated in place, [g_mu | g_alpha] straight into the bf16 operands of the products
            lib.call('gv_iaf_update_bwd_bf16', ptr(z), ptr(net_out[sl]), 2 * d, ptr(colcount[p]), ptr(g_cur),
                     ptr(gld) if p == P - 1 else None, ptr(g_z), ptr(gm_b[L - 1][sl]), gm_b[L - 1].stride(0),
                     ptr(gm_t[L - 1][:, tsl]), gm_t[L - 1].stride(0), ptr(g_old), n, d, st)
            if ctx.chain:                        # g_{l-1} = (g_l W_l) * [a_{l-1} > 0] down to g_x, one launch
                made_chain(gm_b[L - 1][sl], n,
                           [dict(w_packed=wbt[l], n=widths[l - 1], k=widths[l], mask=acts_b[l - 1][sl], out_bf16=gm_b[l - 1][sl],
                                 out_bf16_t=gm_t[l - 1][:, tsl]) for l in reversed(range(1, L))] +
                           [dict(w_packed=wbt[0], n=d, k=widths[0], out_f32=g_old, accumulate=True)], tag='madechain_bwd')
            else:
                for l in reversed(range(1, L)):      # g_{l-1} = (g_l W_l) * [a_{l-1} > 0]
                    gemm_bf16_nt(gm_b[l][sl], wbt[l], n, widths[l - 1], widths[l], mask=acts_b[l - 1][sl], c_bf16=gm_b[l - 1][sl],
                                 c_bf16_t=gm_t[l - 1][:, tsl])
                gemm_bf16_nt(gm_b[0][sl], wbt[0], n, d, widths[0], c_f32=g_old, accumulate=True)
            g_cur = g_old
        # pass 0: the update's gradient w.r.t. the broadcast net row is its column sum; x_old was the zero matrix
        g_row = iaf_bwd_row0(z, acts0[L - 1], colcount[0], g_cur, gld if P == 1 else None, g_z)
        rows0 = [None] * L
        row_gw = row_gb = None
        if ctx.row:     # the row's whole backward chain (masked row gradients, outer products, bias gradients): one launch
            row_gw = [torch.empty(widths[l], ws[l].shape[1], **f32) if ctx.needs_input_grad[3 + l] else None for l in range(L)]
            # a bias whose slice of the optimiser arena is still all-zero takes its gradient there directly
            _verify_direct(ctx)
            direct_b = [t if (t is not None and t.data_ptr() in GRAD_FRESH and t.is_contiguous()) else None for t in ctx.direct_b]
            row_gb = [(direct_b[l] if direct_b[l] is not None else torch.empty(widths[l], **f32))
                      if ctx.has_bias[l] and ctx.needs_input_grad[3 + L + l] else None for l in range(L)]
        else:
            for l in reversed(range(L)):
                rows0[l] = g_row
                if l > 0:
                    mask = acts0[l] if l < L - 1 else None
                    g_row = gemm(g_row, ws[l], a_relu_mask=mask, precision='bf16')
        mtot = S * T * 64 if tiled else max(S, 1) * npad
        # where each layer's bias gradient accumulates, and whether the weight-gradient launch can take it along
        wants_gb = [ctx.has_bias[l] and ctx.needs_input_grad[3 + L + l] for l in range(L)]
        gb_target = [(row_gb[l] if ctx.row else None) if wants_gb[l] else None for l in range(L)]
        fused_gb = [S > 0 and ctx.row and gb_target[l] is not None and ctx.needs_input_grad[3 + l] and row_gw[l] is not None
                    and row_gw[l].stride(0) == ws[l].shape[1]
                    and gemm_bf16_gradw_fits(widths[l], ws[l].shape[1], mtot, max(2, min(GRADW_SPLIT_MAX, mtot // 512))) for l in range(L)]
        if tiled:       # the tiled copies are read by the whole-output product alone: it takes every bias gradient along
            fused_gb = [wants_gb[l] for l in range(L)]
        # The weight-gradient products of this MADE on a SIDE stream (a parallel branch of a captured graph): they depend on
        # nothing later in the backward pass, and the next MADE's backward chains leave half of the CUs idle in their second
        # round of workgroups.  Only when every result goes straight into the optimiser's gradient arena (nothing is handed
        # back to autograd on this stream); the side stream is joined when the whole backward pass has run.
        beside = False
        if tiled and ctx.row and ctx.masks is not None and S > 0:
            _verify_direct(ctx)
            beside = (all(ctx.direct_w[l] is not None and ctx.direct_w[l].data_ptr() in GRAD_FRESH and ctx.direct_w[l].is_contiguous()
                          for l in range(L) if ctx.needs_input_grad[3 + l])
                      and all(direct_b[l] is not None for l in range(L) if wants_gb[l]))
        with backward_side(beside, gm_t, xin_t, acts_t, row_gw, row_gb, g_row, acts0, ws) as on_side:
            ctx.gradw_cap = (GRADW_SPLIT_MAX_SIDE or (128 if len(_made_row_blocks(n)) > 1 else 96)) if on_side else GRADW_SPLIT_MAX
            if ctx.row:
                made_row_bwd(g_row, [dict(w=ws[l], act=acts0[l] if l < L - 1 else None, inp=acts0[l - 1] if l > 0 else None,
                                          gw=row_gw[l], gb=row_gb[l]) for l in range(L)])
            g_ws, g_bs = _MADEForwardBF16._weight_gradients(ctx, L, S, T, tiled, mtot, widths, ws, acts0, rows0, zero_row, row_gw, row_gb,
                                                            direct_b if ctx.row else None, wants_gb, gb_target, fused_gb, gm_t, gm_t_all,
                                                            tb if tiled else 0, xin_t, acts_t, f32, lib.stream())
        return (g_z, None, None, *g_ws, *g_bs, None)

    @staticmethod
    def _weight_gradients(ctx, L, S, T, tiled, mtot, widths, ws, acts0, rows0, zero_row, row_gw, row_gb, direct_b, wants_gb, gb_target,
                          fused_gb, gm_t, gm_t_all, tb, xin_t, acts_t, f32, st):
        """dL/dW_l, dL/db_l of every layer from the stacked passes (+ pass 0's share, already in row_gw / row_gb), the mask fold."""
        g_ws, g_bs, g_bs_acc = [], [], []
        for l in range(L):
            mask0 = acts0[l] if l < L - 1 else None
            inp0 = zero_row if l == 0 else acts0[l - 1]
            gw = gb = None
            if ctx.needs_input_grad[3 + l]:
                gw = row_gw[l] if ctx.row else gemm(rows0[l], inp0, trans_a=True, a_relu_mask=mask0, precision='bf16')
                if S > 0 and not tiled:       # dW_l = g_l^T a_{l-1}: the NT kernel on the transposed copies, reduction over all stacked rows
                    in_t = xin_t if l == 0 else acts_t[l - 1]
                    split = max(2, min(GRADW_SPLIT_MAX, mtot // 512))
                    if fused_gb[l]:      # ... and the stacked passes' share of the bias gradient from the same pass over g_l^T
                        gemm_bf16_gradw(gm_t[l], in_t, widths[l], ws[l].shape[1], mtot, gw, accumulate=True,
                                        a_rowsum=gb_target[l], split_k=split)
                    else:
                        gemm_bf16_nt(gm_t[l], in_t, widths[l], ws[l].shape[1], mtot, c_f32=gw, accumulate=True, split_k=split)
            if ctx.has_bias[l] and ctx.needs_input_grad[3 + L + l]:
                gb = row_gb[l] if ctx.row else colsum(rows0[l], relu_mask=mask0)
            if tiled and (gw is not None or gb is not None):
                # dW_l = g_l^T a_{l-1} over all stacked rows, both operands in 64-row tiles, db_l from the same pass over g_l^T
                gemm_bf16_gradw_tiles(gm_t[l], tb, xin_t if l == 0 else acts_t[l - 1], ctx.t_tile, widths[l], ws[l].shape[1], mtot,
                                      gw if gw is not None else torch.empty(widths[l], ws[l].shape[1], **f32), accumulate=gw is not None,
                                      a_rowsum=gb, split_k=max(2, min(ctx.gradw_cap, mtot // 512)))
            if ctx.has_bias[l] and ctx.needs_input_grad[3 + L + l]:
                if S > 0 and L > 8:
                    rws = torch.empty(int(lib.load().gv_rowsum_bf16_workspace_floats(widths[l], mtot)), **f32)
                    lib.call('gv_rowsum_bf16', ptr(gm_t[l]), gm_t[l].stride(0), widths[l], mtot, ptr(gb), 1, ptr(rws), st)
            g_bs_acc.append(None if fused_gb[l] else gb)      # where the row-sum pass still has to add this layer's share
            if ctx.row and gb is not None and direct_b[l] is not None:
                GRAD_FRESH.discard(gb.data_ptr())
                gb = None
            g_ws.append(gw)
            g_bs.append(gb)
        if S > 0 and L <= 8 and any(b is not None for b in g_bs_acc):
            # the stacked passes' share of every bias gradient: ONE pass over the transposed gradient buffers of all layers
            rws = torch.empty(int(lib.load().gv_rowsum_bf16_workspace_floats(sum(widths), mtot)), **f32)
            outs = (_ct.c_void_p * L)(*[ptr(b) for b in g_bs_acc])
            segs = (_ct.c_int32 * L)(*widths)
            lib.call('gv_rowsum_bf16_segments', ptr(gm_t_all), gm_t_all.stride(0), sum(widths), mtot, L, _ct.addressof(outs),
                     _ct.addressof(segs), 1, ptr(rws), st)
        if ctx.masks is not None:       # dL/dW = mask * dL/d(mask * W): one launch for all layers, straight into the arena where fresh
            _verify_direct(ctx)
            idx = [l for l in range(L) if g_ws[l] is not None]
            tgt = [ctx.direct_w[l] if (ctx.direct_w[l] is not None and ctx.direct_w[l].data_ptr() in GRAD_FRESH
                                       and ctx.direct_w[l].is_contiguous()) else None for l in idx]
            res = mul_multi([ctx.masks[l] for l in idx], [g_ws[l] for l in idx], outs=tgt) if idx else []
            for l, t, r in zip(idx, tgt, res):
                if t is not None:
                    GRAD_FRESH.discard(t.data_ptr())
                    g_ws[l] = None
                else:
                    g_ws[l] = r
        return g_ws, g_bs


MADE_BF16_STORAGE = _os.environ.get('GV_MADE_BF16', '1') == '1'
GRADW_SPLIT_MAX = int(_os.environ.get('GV_GRADW_SPLIT_MAX', '256'))      # most K slices of a MADE weight-gradient product
# ... when the products run on the side stream, beside the next flow's backward chains: a slice is a workgroup that takes a whole
# CU's LDS, so fewer of them leave the chains more of the chip, and the partial sums are fewer (alone 256 slices are fastest: 58 us
# against 73 at 128; in the c3 step 5.79 ms at 256, 5.72-5.77 at 128, 5.67-5.69 at 96, 5.72 at 64).  0 = by size: 96, and 128 where
# the passes run over two row blocks -- since the backward chains run two workgroups per CU (GV_CHAIN_WIDE0) the c3 step is 4.98-5.01 ms
# at 128 against 5.06-5.08 at 96; FB15k-237 size + 3 IAF blocks stays best at 96 (2.83 against 2.86)
GRADW_SPLIT_MAX_SIDE = int(_os.environ.get('GV_GRADW_SPLIT_MAX_SIDE', '0'))
MADE_CHAIN_IAF = _os.environ.get('GV_MADE_CHAIN_IAF', '1') == '1'      # the IAF update inside the chain's last layer
# ... and that many passes of the backward per launch (gv_chain_iafb.n_passes).  The looped launch needs 28 % less kernel time per
# pass (FB15k-237 size, 3 IAF blocks: 3 x 203 us against 15 x 56 us) and LOSES as a step (2.96 -> 3.12 ms; WN18RR 5.24 -> 5.44): its
# workgroups hold their CU's LDS (134 KB) through all passes, and the weight-gradient products of the block before, which run beside
# it on the side stream and are the longer of the two, only get the CUs the chain leaves free (k_gemm_bf16_tallk: 50 -> 63 us).
# tools/probes/passes_grid.sh: no split / row-block setting turns that around.  1 = a launch per pass.
MADE_CHAIN_PASSES = max(1, min(6, int(_os.environ.get('GV_MADE_CHAIN_PASSES', '1'))))
# passes of the forward per gv_made_chain_fwd launch (1: a gv_made_chain launch per pass).  0 = by size: all of them (up to six) where
# a pass's row tiles fit the chip's workgroup slots (FB15k-237 size + 3 IAF blocks: 2.96 -> 2.80 ms), three where they do not and the
# passes run over two row blocks (WN18RR, 640 tiles on 512 slots: the second round of workgroups lasts as long as a launch does --
# 5.18 ms per pass-launch, 5.05 / 4.99 / 5.11 with two / three / all passes per launch)
MADE_FWD_PASSES = max(0, min(6, int(_os.environ.get('GV_MADE_FWD_PASSES', '0'))))
MADE_FWD_ROW0 = _os.environ.get('GV_MADE_FWD_ROW0', '1') == '1'      # ... with pass 0's update as that launch's first stage
MADE_CHAIN_IAFB = _os.environ.get('GV_MADE_CHAIN_IAFB', '1') == '1'    # ... and its backward as the backward chain's first stage
MADE_T_TILES = _os.environ.get('GV_MADE_T_TILES', '1') == '1'          # ... and the transposed copies in tiles of 64 rows


MADE_ROW_BLOCKS = int(_os.environ.get('GV_MADE_ROW_BLOCKS', '2'))       # independent row blocks of a MADE's passes (1: off)
# ... of the fp32 node (a launch per product: 9.47 -> 9.33 ms for the mini-batch step with 3 IAF blocks, 9.52 -> 9.32 on the full
# FB15k-237-sized graph; three blocks 10.4), from 128 row tiles on
MADE_F32_ROW_BLOCKS = int(_os.environ.get('GV_MADE_F32_ROW_BLOCKS', '2'))
MADE_F32_ROW_BLOCKS_MIN_TILES = int(_os.environ.get('GV_MADE_F32_ROW_BLOCKS_MIN_TILES', '128'))
MADE_ROW_BLOCKS_MIN_TILES = int(_os.environ.get('GV_MADE_ROW_BLOCKS_MIN_TILES', '0'))     # 0: more row tiles than chain workgroups fit the chip
_chain_slots = {}


def _made_row_blocks(n, want=None, min_tiles=None):
    """Row ranges a MADE's passes are run over, as independent launch sequences on their own streams.  Every launch of a pass is
    row-local (a chain workgroup owns 64 rows through all layers, the update and its backward are element-wise), so the passes of
    one row block depend on nothing in another block -- but as ONE sequence of launches every pass waits for the last workgroup
    of the one before it.  That is expensive here: two chain workgroups fit a CU, so the 640 workgroups of a WN18RR pass run a
    second round on a quarter of the chip (forward chain: 60 us at 512 row tiles, 88 us at 576), and the chain (latency-bound,
    ~2 TB/s) alternates with the update's backward (bandwidth-bound) instead of running beside it.  Two blocks of 320 row tiles
    on two streams: 6.32 -> 5.72 ms per step (blocks cut at the last full round of workgroups, 512 + 128 tiles: 6.01; three
    blocks 6.08, four 6.46).  Only where a pass has more row tiles than the chip holds chain workgroups: at FB15k-237 size (228
    tiles, one partial round) two blocks cost 3.36 -> 3.42 ms."""
    want = MADE_ROW_BLOCKS if want is None else want
    if want <= 1:
        return [(0, n)]
    tiles = (n + 63) // 64
    least = MADE_ROW_BLOCKS_MIN_TILES if min_tiles is None else min_tiles
    if least <= 0:
        dev = torch.cuda.current_device()
        if dev not in _chain_slots:
            _chain_slots[dev] = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
        least = _chain_slots[dev] + 1
    k = want if tiles >= least else 1
    k = max(1, min(k, tiles))
    cuts = [(tiles * i // k) * 64 for i in range(k)] + [n]
    return [(cuts[i], cuts[i + 1]) for i in range(k)]


# ONE fork and ONE join of the row blocks per flow STACK (forward): everything between two bf16 MADE nodes of an IAF stack is row-local
# once the PermuteLayer and pass 0's update ride in the nodes' launches, so a block's launches of node k + 1 depend on the same block's
# launches of node k alone -- the side streams stay forked from the first node to the last, the log-det row sums (the only launches
# over all rows) run behind the one join.  A cross-stream edge costs a replayed graph 20-40 us; three IAF blocks had six per direction.
# Temporaries of the nodes are kept until the join (a freed buffer could otherwise be handed to another block's stream).
_FORK = None
MADE_KEEP_FORKED = _os.environ.get('GV_MADE_KEEP_FORKED', '1') == '1'


@contextlib.contextmanager
def keep_row_blocks_forked(n):
    """Around the MADE forward calls of one flow stack on n rows (encoders.KGVAE._apply_flows)."""
    global _FORK
    layout = _ops.launch_layout()
    blocks = _made_row_blocks(n) if layout.row_blocks else [(0, n)]
    if len(blocks) == 1 or layout.process_group or not MADE_KEEP_FORKED or _FORK is not None or _ops.GEMM_PRECISION != 'bf16':
        yield
        return
    join_prepared()                           # (parameter-only work of every node: joined once, in front of the fork)
    main = torch.cuda.current_stream()
    sides = [_side(('made_rows', i)) for i in range(1, len(blocks))]
    for sd in sides:
        sd.wait_stream(main)
    _FORK = dict(n=n, blocks=blocks, sides=sides, hold=[], deferred=[], main=main, open=True)
    try:
        yield
    finally:
        st, _FORK = _FORK, None
        _fork_join(st)
        for fn in st['deferred']:
            fn()
        st['hold'].clear()


def _fork_join(st):
    if st['open']:
        for sd in st['sides']:
            st['main'].wait_stream(sd)
        st['open'] = False


def fork_sync():
    """Something that is not row-blocked is about to read what the blocks wrote: join now (and fork again behind it)."""
    if _FORK is not None and _FORK['open']:
        _fork_join(_FORK)


def _fork_reopen():
    if _FORK is not None and not _FORK['open']:
        for sd in _FORK['sides']:
            sd.wait_stream(_FORK['main'])
        _FORK['open'] = True


def join_prepared():
    """Wait (once) for the parameter-only work made_prepare left on its side stream."""
    for prep in _made_prep.values():
        if not prep['joined'][0]:
            torch.cuda.current_stream().wait_event(prep['done'])
            prep['joined'][0] = True
        break


def _by_row_blocks(run, n, want, min_tiles=None, keep_forked=False):
    """run(r0, r1) over the row blocks of _made_row_blocks (want: how many, None: MADE_ROW_BLOCKS; <= 1: all rows at once): the
    first on the current stream, the others on side streams that are joined before returning (under hipGraph capture: parallel
    branches)."""
    if keep_forked and _FORK is not None and _FORK['n'] == n and want is None and min_tiles is None:
        _fork_reopen()
        run(*_FORK['blocks'][0])
        for sd, blk in zip(_FORK['sides'], _FORK['blocks'][1:]):
            with torch.cuda.stream(sd):
                run(*blk)
        return
    fork_sync()
    blocks = _made_row_blocks(n, want, min_tiles) if _ops.launch_layout().row_blocks else [(0, n)]      # (timed launches are whole launches: bench.py's K4 line)
    if len(blocks) == 1:
        run(0, n)
        return
    main = torch.cuda.current_stream()
    sides = [_side(('made_rows', i)) for i in range(1, len(blocks))]
    for sd in sides:
        sd.wait_stream(main)
    run(*blocks[0])
    for sd, blk in zip(sides, blocks[1:]):
        with torch.cuda.stream(sd):
            run(*blk)
    for sd in sides:
        main.wait_stream(sd)


def made_forward(z, colcount, weights, biases, masks=None, reverse_out=False):
    """MADE.forward as one autograd node; with bf16 dense products (set_gemm_precision('bf16'), BASELINE configs[2]) and
    layer widths that are multiples of 8 the bf16-storage pipeline of csrc/k_made.hip runs.  ``masks``: the autoregressive
    masks when ``weights`` are the RAW parameters (MaskedLinear.weight); None when the masks are folded in already."""
    if (_ops.GEMM_PRECISION == 'bf16' and MADE_BF16_STORAGE and z.shape[1] % 8 == 0 and all(w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0
                                                                                       for w in weights)):
        if masks is not None and len(weights) > 8:           # gv_mul_multi's table holds 8 entries
            weights, masks = [masked_weight(m, w) for m, w in zip(masks, weights)], None
        return _MADEForwardBF16.apply(z, colcount, tuple(masks) if masks is not None else None, *weights, *biases, bool(reverse_out))
    if masks is not None and len(weights) > 8:               # gv_mul_multi's table holds 8 entries
        weights, masks = [masked_weight(m, w) for m, w in zip(masks, weights)], None
    return _MADEForward.apply(z, colcount, tuple(masks) if masks is not None else None, bool(reverse_out), *weights, *biases)
