"""Tensor-level wrappers over the C ABI: index builders, raw ops, and torch.autograd Functions.

Every function here requires CUDA (ROCm) float32 tensors and the built HIP library; nothing
falls back to torch CPU math.  torch is used for device memory, the current stream and autograd
bookkeeping only (plus sort/cumsum when an index is built, outside the hot step).
"""
import contextlib
import ctypes as _ct
import itertools as _it
import os as _os
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib
from .lib import ptr

ACT_NONE, ACT_RELU = 0, 1

# Direct gradient targets: parameter storage address -> the gradient buffer its gradient should be ADDED
# into (registered by optim.FlatAdam, whose arena is zeroed once per step).  A backward that finds its
# parameter here accumulates in place and returns None for it, so autograd launches no zero-fill and no
# AccumulateGrad add for that parameter.  Empty registry = ordinary autograd behaviour.
DIRECT_GRAD = {}
# Gradient buffers (keyed by address) that are known to be all-zero right now: the optimiser adds its arena slices after
# zero_grad() / step(); the first backward kernel that writes one removes it.  Only a buffer listed here may be
# OVERWRITTEN by a backward kernel (the identity-embedding path below); anything else is accumulated into.
GRAD_FRESH = set()
# Bumped whenever the registry changes (an optimiser is built or dropped).  A Function that resolved a direct target in its
# forward stamps the value on its ctx; its backward refuses to run if the registry changed in between (it would add into an
# arena that is no longer the parameters' -- INTEGRATION.md, "The optimiser contract").
DIRECT_EPOCH = [0]


def _stamp_direct(ctx):
    ctx._gv_direct_epoch = DIRECT_EPOCH[0]


def _verify_direct(ctx):
    if getattr(ctx, '_gv_direct_epoch', DIRECT_EPOCH[0]) != DIRECT_EPOCH[0]:
        raise RuntimeError('the optimiser gradient arena (ops.DIRECT_GRAD) changed between this forward and its backward: '
                           'build / drop FlatAdam outside a forward-backward pair')


def _direct(t):
    tgt = DIRECT_GRAD.get(t.data_ptr()) if t is not None else None
    return tgt if (tgt is not None and tgt.shape == t.shape) else None


def _direct_flat(t):
    """Direct target of a contiguous VIEW of a parameter that starts at its storage (e.g. z_pre.squeeze(0))."""
    tgt = DIRECT_GRAD.get(t.data_ptr()) if t is not None else None
    return tgt.view(t.shape) if (tgt is not None and tgt.numel() == t.numel() and tgt.is_contiguous()) else None
EPILOGUE_COLSUM_SLICES = 1024  # = GV_EPILOGUE_COLSUM_SLICES (include/gcnvae.h)
DEFAULT_CHUNK = 256        # max edges per work item of the dst/src-sorted aggregations
DEFAULT_CHUNK_REL = 128    # max edges per work item of the by-relation weight gradient


CHUNK_DIV = int(_os.environ.get('GV_CHUNK_DIV', '4096'))


def chunk_for(n_entries: int, most: int = DEFAULT_CHUNK) -> int:
    """Edges per work item for a list of ``n_entries``.  One wave walks an item's edges in order (four in flight), so on a SMALL
    graph the longest item, not the edge count, sets a launch's duration: a sampled batch of 20 000 edges whose hub rows were
    cut into 256-edge items spent 40-80 us per aggregation on 64 serial steps of one wave.  Items shrink with the list: a power of
    two, n / 4096 rounded down, between 16 and ``most`` (measured over n / 2048, 4096, 8192: FB15k-237-sized graphs -- 0.54 M
    edges -- run best on 128-edge items: 1.049 vs 1.058 ms at h = 200, 3.46 vs 3.56 ms at h = 500; lists of >= 1 M entries keep 256)."""
    env = _os.environ.get('GV_CHUNK')
    if env:
        return max(1, int(env))
    # by-relation lists (most = DEFAULT_CHUNK_REL) keep larger items: few, long segments -- 22 relations of 7 900 edges at WN18RR
    # size -- pay for every extra partial slot in the fix-up (weight-gradient launches 58 / 68 us with 128-edge items, 67 / 92 with 32)
    div = CHUNK_DIV // 4 if most == DEFAULT_CHUNK_REL else CHUNK_DIV
    c = 16
    while c * 2 <= most and c * 2 * div <= int(n_entries):
        c *= 2
    return c
DIST_FWD_CHUNKS = 2        # destination-row blocks whose all-reduce overlaps the next block's aggregation


# ------------------------------------------------------------------------------------------------
# fork / join on side streams: independent groups of small, latency-bound launches (KL, MMD, scalar
# reductions, weight-gradient GEMMs) run beside the bandwidth-bound kernels instead of behind them.
# Under hipGraph capture the event edges become parallel graph branches.  Buffers a branch writes are
# allocated by the caller BEFORE the fork (torch's allocator is per stream).  Opt in with GV_CONCURRENCY=1.
CONCURRENCY = _os.environ.get('GV_CONCURRENCY', '0') == '1'   # measured null on MI355X (1.49 vs 1.49 ms/step): off by default
_side_streams = {}


def _side(i):
    key = (torch.cuda.current_device(), i)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream()
    return _side_streams[key]


@contextlib.contextmanager
def fork(i):
    """Run the enclosed launches on side stream ``i`` (after everything already enqueued on the current
    stream); pair with ``join(i)`` before their results are consumed."""
    if not CONCURRENCY:
        yield
        return
    s = _side(i)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        yield


def join(*ids):
    if CONCURRENCY:
        cur = torch.cuda.current_stream()
        for i in ids:
            cur.wait_stream(_side(i))


# Parameter gradients BESIDE the backward pass.  A launch whose only consumer is the optimiser (a weight gradient written straight
# into the gradient arena) need not hold up the kernels that feed autograd: it goes to one side stream -- a parallel branch of a
# captured graph -- and that stream is joined once, when the engine has run the whole backward pass (the optimiser step comes after).
# What such launches read is kept from the allocator until the join (autograd frees a node's saved tensors as soon as the node ran).
# Off when a process group exists: the gradients then feed all-reduces started inside the backward pass, and a captured step is a
# chain of graph segments cut at the collectives (a stream forked in one segment cannot be joined in another).
# Used by the MADE backward (weight-gradient products that need one workgroup per CU, beside backward chains whose second round
# of workgroups leaves half of the CUs idle: -0.13 ms at WN18RR size) and, since round 4, by the R-GCN layers' weight gradients
# where the layer is large (RGCN_BWD_SIDE below; in round 3, with the gradients going through AccumulateGrad and other kernels,
# the same cost more than it hid: FB15k-237 1.047 -> 1.089 ms per step, the mini-batch step 0.98 -> 1.11).  Not by the decoder's.
BWD_SIDE = _os.environ.get('GV_BWD_SIDE', '1') == '1'
_bwd_side_held = []
# ... the R-GCN layers' weight gradients too (round 4, with the arena targets): 'auto' = where the layer is large and the side stream
# carries nothing else in this backward pass -- measured per step: h = 500 3.41 -> 3.28 ms, 1 M nodes / 50 M edges 81.4 -> 78.9,
# FB15k-237 size 1.026 -> 1.020; NOT the mini-batch graph (20 000 edges: 0.967 -> 1.006) and not behind IAF blocks, whose weight-gradient
# products already run there (WN18RR + 3 IAF 4.98 -> 5.19, mini-batch + 3 IAF fp32 5.03 -> 5.15).  '0' / '1': never / wherever possible.
# (The decoder's relation-side gradient the same way: 1.020 -> 1.048 ms -- a 24-us launch does not carry its fork and join.)
RGCN_BWD_SIDE = _os.environ.get('GV_RGCN_BWD_SIDE', 'auto')
RGCN_BWD_SIDE_MIN_WORK = 10 ** 8       # edges x widest side of the layer


_bwd_side_others = [False]     # something other than an R-GCN layer put work on the side stream in this backward pass


def rgcn_bwd_side(num_edges, width):
    layout = launch_layout()
    if not layout.bwd_side or layout.process_group:      # (under a process group the arena feeds collectives started inside the pass)
        return False
    if RGCN_BWD_SIDE == 'auto':
        return not _bwd_side_others[0] and num_edges * width >= RGCN_BWD_SIDE_MIN_WORK
    return RGCN_BWD_SIDE == '1'


@dataclass(frozen=True)
class LaunchLayout:
    """How a step may spread its launches over streams -- decided in ONE place (``launch_layout()``) from the process state, read by
    ``backward_side``, ``made.made_prepare`` and ``made._by_row_blocks``."""
    timed: bool             # a lib.KernelTimer is installed (bench.py's per-kernel figures): every launch runs alone, in order
    process_group: bool     # a torch.distributed group exists: the step carries collectives, a captured step is a chain of segments
    bwd_side: bool          # a MADE's weight-gradient products beside the rest of the backward pass (side stream 'bwd')
    bwd_side_join_at_collective: bool   # ... joined in front of the next collective (a fork must not outlive its graph segment)
    made_prepare: bool      # the flows' parameter-only work beside the encoder's layers (side stream 'made_prep')
    row_blocks: bool        # a MADE's passes over independent row blocks on their own streams (forked and joined inside the node)


def launch_layout():
    from . import made as _made          # (made imports ops: resolved at call time)
    timed, pg = lib.TIMER is not None, _process_group()
    return LaunchLayout(
        timed=timed, process_group=pg,
        # the side stream is joined when the backward pass has run -- or, with collectives on the step, in front of the first
        # collective after the fork (distributed.start_collective -> backward_side_finish): the products then still run beside the
        # remaining backward chains of the flow stack, which is where they were hidden anyway
        bwd_side=BWD_SIDE and not timed, bwd_side_join_at_collective=pg,
        # forked at the START of the forward pass, in front of the encoder's collectives: not under a process group
        made_prepare=_made.MADE_PREPARE and not timed and not pg,
        row_blocks=not timed)


def _process_group():
    """A torch.distributed process group exists (even of one rank): the step then carries collectives, a captured step is a chain
    of graph SEGMENTS cut at them, and a stream forked in one segment may not be joined in another -- no deferred side streams."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


@contextlib.contextmanager
def backward_side(enabled, *held, rgcn=False):
    """Inside an autograd backward: run the enclosed launches on the side stream (after everything already enqueued); ``held``:
    the tensors they touch.  Yields whether the side stream is in use."""
    if not (enabled and launch_layout().bwd_side):      # (timed launches run alone: bench.py's per-kernel lines)
        # A node that stays on the main stream while an EARLIER node of this backward pass left work on the side stream: the two
        # may write the same slices of the gradient arena (one MADE's parameters behind two nodes -- the posterior pass and a
        # separate MMD prior pass; the first stores on the side stream, the second accumulates, or hands its gradient to
        # AccumulateGrad, on this one).  Order them: this stream waits for the side stream first.
        if _bwd_side_held:
            torch.cuda.current_stream().wait_stream(_side('bwd'))
        yield False
        return
    side, main = _side('bwd'), torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        yield True
    first = not _bwd_side_held
    _bwd_side_held.append(held)
    if not rgcn:
        _bwd_side_others[0] = True
    if first:
        def _join():
            if not _bwd_side_held:      # joined already (in front of a collective: backward_side_finish)
                return
            # the stream the pass was started under AND the one current now (backward() may be called under another stream than
            # the node's launches ran on; the optimiser reads the arena on the caller's)
            main.wait_stream(side)
            cur = torch.cuda.current_stream()
            if cur != main:
                cur.wait_stream(side)
            _bwd_side_held.clear()
            _bwd_side_others[0] = False
        torch.autograd.Variable._execution_engine.queue_callback(_join)


def backward_side_finish():
    """Join the side stream if a backward pass left work on it without reaching its end (an exception inside the pass): called
    by the optimiser before it reads or clears the gradient arena."""
    if _bwd_side_held:
        torch.cuda.current_stream().wait_stream(_side('bwd'))
        _bwd_side_held.clear()
    # unconditionally: a pass that raised (or ran without the engine's end-of-pass callback) after a MADE node had set the flag would
    # otherwise keep rgcn_bwd_side('auto') off for the rest of the process
    _bwd_side_others[0] = False


class StayOnDevice:
    """Mixin of the HIP modules: once their parameters live on a ROCm device, ``module.cpu()`` / ``.to('cpu')`` leaves them
    there.  The reference moves its model to the host for validation (kgvae/link_predict.py:239-242, "full graph is too
    large") and back; there is no CPU path here to move to -- the compute stays on the GPU and ``to_module_device`` brings
    whatever the caller hands in (node ids, relation types, norms on the host) to the parameters.  Not a fallback."""
    _gv_warned = False

    def _apply(self, fn, *args, **kwargs):
        dev = next((t.device for t in list(self.parameters()) + list(self.buffers()) if t.is_cuda), None)
        if dev is not None:
            try:
                moved = fn(torch.zeros(1, device=dev))
            except Exception:          # a fn that does not take a plain tensor: not a device move
                moved = None
            if moved is not None and moved.device.type == 'cpu':
                if not StayOnDevice._gv_warned:
                    StayOnDevice._gv_warned = True
                    import sys
                    print('[gcn_vae_amd] .cpu() on a HIP module is a no-op: parameters stay on %s, inputs are copied to them'
                          % dev, file=sys.stderr)
                return self
        return super()._apply(fn, *args, **kwargs)


def to_module_device(param, *tensors):
    """The caller's tensors on the device of ``param`` (a host -> device copy when they are elsewhere; None passes through)."""
    dev = param.device
    out = tuple(t if (t is None or not isinstance(t, torch.Tensor) or t.device == dev) else t.to(dev) for t in tensors)
    return out if len(out) != 1 else out[0]


def _chk(t, dtype=torch.float32, name='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name}: the gfx950 path needs a CUDA/ROCm tensor (got '
                           f'{type(t).__name__} on {getattr(t, "device", "?")}); there is no CPU fallback')
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')
    return t


def _row_major(t, name='matrix'):
    """2-D fp32 CUDA tensor with unit inner stride; returns (tensor, leading dimension)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f'{name}: the gfx950 path needs a CUDA/ROCm tensor; there is no CPU fallback')
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f'{name}: expected 2-D float32, got {t.dtype} {tuple(t.shape)}')
    if t.stride(1) != 1 or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t.shape[1])


# ------------------------------------------------------------------------------------------------
# work-item lists, graph / relation / triplet indices, the phase and LDS-resident K1 launch forms: indices.py (same namespace
# for the callers: ops.GraphIndex, ops.bdd_aggregate_phases, ...)
# ------------------------------------------------------------------------------------------------
from .indices import *                                           # noqa: E402,F401,F403
from .indices import _carve_i32, _index_caps, _index_workspace, _rowptr_from_sorted      # noqa: E402,F401
from . import indices                                            # noqa: E402,F401  (ops.indices.<KNOB>: where that file's knobs live)
from . import batch_index                                        # noqa: E402,F401
from .batch_index import build_batch_indices                      # noqa: E402,F401


# ------------------------------------------------------------------------------------------------
# raw ops
RGCN_SIDE_GRADW_FIRST = _os.environ.get('GV_RGCN_SIDE_GRADW_FIRST', 'auto')


def rgcn_side_gradw_first(num_edges, width):
    """Order of the layer's two weight gradients on the backward side stream.  The main stream runs the self-loop's dL/dx product
    (MFMA-bound) and then the K1^T aggregation (cache-bandwidth-bound); with the relation-weight gradient (cache-bandwidth-bound)
    FIRST on the side stream and the self-loop's weight-gradient product (MFMA-bound) behind it, the launches that run beside each
    other want different things.  Measured: FB15k-237 size, h = 200: 1.005 -> 0.986 ms; h = 500: 2.993 -> 2.999; 1 M nodes / 50 M
    edges: 75.0 -> 75.2; WN18RR + 3 IAF: the same.  'auto': cache-resident graphs at input widths up to 256; '0' / '1' force."""
    if RGCN_SIDE_GRADW_FIRST in ('0', '1'):
        return RGCN_SIDE_GRADW_FIRST == '1'
    return num_edges <= 4_000_000 and width <= 256


def _k1_items(gidx, seg):
    """The work-item list a single-GPU K1 launch over a static graph walks: long items first where that is switched on."""
    if indices.K1_ITEMS_LARGEST_FIRST and not gidx.sync_free:
        return indices.largest_first(seg)
    return seg


def pack_supported(num_bases, blk_in, blk_out, transpose_w=False):
    return bool(lib.load().gv_rgcn_bdd_pack_supported(num_bases, blk_in, blk_out, 1 if transpose_w else 0))


def pack_weight(weight, num_bases, blk_in, blk_out, transpose_w=False):
    """Lane-packed copy of a bdd relation-weight matrix for one K1 launch kind (see include/gcnvae.h)."""
    weight = _chk(weight, name='weight')
    packed = torch.empty_like(weight)
    lib.call('gv_rgcn_bdd_pack_weight', ptr(weight), weight.shape[0], num_bases, blk_in, blk_out,
             1 if transpose_w else 0, ptr(packed), lib.stream())
    return packed


def _check_items(seg: SegmentItems):
    """The kernels index these lists by launch geometry: refuse a handle whose counts exceed its buffers."""
    if seg.n_items < 0 or seg.n_fix < 0 or seg.items.numel() < 4 * seg.n_items or seg.fix.numel() < 4 * seg.n_fix:
        raise ValueError(f'inconsistent work-item lists: n_items={seg.n_items} (buffer {seg.items.numel() // 4}), '
                         f'n_fix={seg.n_fix} (buffer {seg.fix.numel() // 4})')
    if seg.n_items > 0 and seg.items.data_ptr() == 0:
        raise ValueError('work-item list has no storage')


def bdd_aggregate(seg: SegmentItems, nbr, etype, coef, coef_idx, feat, weight, num_bases, blk_in, blk_out,
                  transpose_w=False, addend=None, act=ACT_NONE, keep=None, keep_scale=1.0, out=None, packed=False):
    feat, ld_feat = _row_major(feat, 'feat')
    weight = _chk(weight, name='weight')
    n_seg = seg.rowptr.numel() - 1
    out_dim = num_bases * blk_out
    if feat.shape[1] != num_bases * blk_in:
        raise ValueError(f'feat has {feat.shape[1]} columns, expected num_bases*blk_in = {num_bases * blk_in}')
    num_rels = weight.shape[0]
    if weight.numel() != num_rels * num_bases * blk_in * blk_out:
        raise ValueError('weight shape does not match (R, num_bases*blk_in*blk_out)')
    if out is None:
        out = torch.empty(n_seg, out_dim, dtype=torch.float32, device=feat.device)
    ld_add = 0
    if addend is not None:
        addend, ld_add = _row_major(addend, 'addend')
        if tuple(addend.shape) != (n_seg, out_dim):
            raise ValueError('addend shape mismatch')
    if keep is not None:
        _chk(keep, torch.uint8, 'keep')
        if tuple(keep.shape) != (n_seg, out_dim):
            raise ValueError('keep shape mismatch')
    if coef is not None:
        coef = _chk(coef.reshape(-1), name='coef')
    _check_items(seg)
    partial = None
    if seg.n_fix > 0:
        partial = torch.empty(seg.n_slots, out_dim, dtype=torch.float32, device=feat.device)
    ld_out = out.stride(0) if n_seg > 1 else out_dim
    tag = f'agg_{"T" if transpose_w else "N"}_{blk_in}x{blk_out}_nb{num_bases}'
    # (per-kernel timing, bench.py: the aggregation kernel is timed alone, the split rows' fix-up launched behind it)
    timed = lib.TIMER is not None
    lib.call('gv_rgcn_bdd_aggregate', ptr(seg.items), seg.n_items, ptr(seg.fix), 0 if timed else seg.n_fix, ptr(nbr),
             ptr(etype), ptr(coef), ptr(coef_idx), ptr(feat), ld_feat, ptr(weight), num_rels, num_bases, blk_in,
             blk_out, 1 if transpose_w else 0, 1 if packed else 0, ptr(addend), ld_add, act, ptr(keep),
             float(keep_scale), ptr(out), ld_out, ptr(partial), lib.stream(), tag=tag)
    if timed and seg.n_fix > 0:      # the aggregation kernel was timed alone; finish the split rows
        lib.call('gv_rgcn_bdd_fixup', ptr(seg.fix), seg.n_fix, ptr(partial), out_dim, ptr(addend), ld_add, act,
                 ptr(keep), float(keep_scale), ptr(out), ld_out, lib.stream())
    return out


def bdd_grad_weight(seg: SegmentItems, src, dst, coef, coef_idx, x, g, num_bases, blk_in, blk_out, out=None,
                    accumulate=False):
    x, ld_x = _row_major(x, 'x')
    g, ld_g = _row_major(g, 'g')
    n_seg = seg.rowptr.numel() - 1
    w_row = num_bases * blk_in * blk_out
    if out is None:
        out = torch.empty(n_seg, w_row, dtype=torch.float32, device=x.device)
        accumulate = False
    if coef is not None:
        coef = _chk(coef.reshape(-1), name='coef')
    partial = None
    if seg.n_fix > 0:
        partial = torch.empty(seg.n_slots, w_row, dtype=torch.float32, device=x.device)
    lib.call('gv_rgcn_bdd_grad_weight', ptr(seg.items), seg.n_items, ptr(seg.fix), seg.n_fix, ptr(src), ptr(dst),
             ptr(coef), ptr(coef_idx), ptr(x), ld_x, ptr(g), ld_g, num_bases, blk_in, blk_out, ptr(out), ptr(partial),
             1 if accumulate else 0, lib.stream(), tag=f'gradw_{blk_in}x{blk_out}_nb{num_bases}')
    return out


GEMM_PRECISION = 'f32'      # 'f32': exact fp32 MFMA; 'bf16': bf16 operands, fp32 accumulation (BASELINE configs[2])


def set_gemm_precision(precision):
    """Precision of every dense product on the path (MaskedLinear, self-loop term, evaluation scorer)."""
    global GEMM_PRECISION
    if precision not in ('f32', 'bf16'):
        raise ValueError("precision must be 'f32' or 'bf16'")
    GEMM_PRECISION = precision


@contextlib.contextmanager
def gemm_precision(precision):
    old = GEMM_PRECISION
    set_gemm_precision(precision)
    try:
        yield
    finally:
        set_gemm_precision(old)


LIVE_ROWS = None        # (device int32 (1,), cap): inside a static-shape batch step, node arrays of cap rows hold *rows_dev real rows


class live_rows:
    """``with ops.live_rows(rows_dev, cap):`` -- the fp32 products over node arrays of exactly ``cap`` rows skip the padding rows
    (gv_gemm_f32_live_rows: zero rows out, a shorter reduction for the weight gradients; gv_made_chain_f32: workgroups that hold
    only padding rows).  graph_step.GraphedMiniBatchStep wraps the step in it: ~30 % of a sampled batch's 14 541 padded rows are
    padding.
    CONTRACT: inside the context EVERY fp32 operand of exactly ``cap`` rows is taken for a padded node array -- the match is by
    row count, nothing marks the arrays.  The caller must not run products over other (cap, k) operands inside it (the captured
    mini-batch step does not: its only other row counts are cap + 200 and 200).  The VALUES of padding rows behind the first
    ``*rows_dev`` are unspecified (zeros from skipped tiles / workgroups, act(bias) from tiles that straddle the boundary); they
    take no part in any sum, mean or gradient."""

    def __init__(self, rows_dev, cap):
        self.val = (rows_dev, int(cap)) if rows_dev is not None else None

    def __enter__(self):
        global LIVE_ROWS
        self.saved, LIVE_ROWS = LIVE_ROWS, self.val
        return self

    def __exit__(self, *exc):
        global LIVE_ROWS
        LIVE_ROWS = self.saved
        return False


def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=ACT_NONE, out=None, accumulate=False, split_k=1,
         a_relu_mask=None, precision=None, b_k_chunks=None, c_tiles=None):
    """out = act(op(a') @ op(b) + bias) (+ out);  a' = a * [a_relu_mask > 0] when a mask (same layout as a) is given.
    precision None = the global GEMM_PRECISION.  b_k_chunks / c_tiles (int64 device words, ops.block_words): blocks of op(b) / tiles
    of the result known to be zero are skipped (gv_gemm_f32_sparse; fp32 only)."""
    a, lda = _row_major(a, 'a')
    b, ldb = _row_major(b, 'b')
    m, k = (a.shape[1], a.shape[0]) if trans_a else a.shape
    kb, n = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if k != kb:
        raise ValueError(f'gemm inner dimensions differ: {k} vs {kb}')
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
        accumulate = False
    if bias is not None:
        _chk(bias, name='bias')
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = int(lib.load().gv_gemm_workspace_bytes(m, n, k, split_k))
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=a.device)
    if a_relu_mask is not None:
        a_relu_mask, ld_mask = _row_major(a_relu_mask, 'a_relu_mask')
        if tuple(a_relu_mask.shape) != tuple(a.shape) or ld_mask != lda:
            raise ValueError('a_relu_mask must have the shape and leading dimension of a')
    entry = 'gv_gemm_bf16' if (precision or GEMM_PRECISION) == 'bf16' else 'gv_gemm_f32'
    live = LIVE_ROWS
    if (b_k_chunks is not None or c_tiles is not None) and entry == 'gv_gemm_f32':
        if b_k_chunks is not None and (b_k_chunks.dtype != torch.int64 or b_k_chunks.numel() < (n + 63) // 64 or k > 1024 or split_k != 1):
            raise ValueError('b_k_chunks: one int64 word per 64-column tile, k <= 1024, no split-K')
        if c_tiles is not None and (c_tiles.dtype != torch.int64 or c_tiles.numel() * 64 < ((m + 63) // 64) * ((n + 63) // 64)):
            raise ValueError('c_tiles: one bit per 64 x 64 tile of the result')
        rows = live[0] if (live is not None and a.shape[0] == live[1] and a.device == live[0].device) else None
        lib.call('gv_gemm_f32_sparse', 1 if trans_a else 0, 1 if trans_b else 0, m, n, k, ptr(a), lda, ptr(b), ldb, ptr(out),
                 out.stride(0) if m > 1 else n, ptr(bias), act, 1 if accumulate else 0, split_k, ptr(a_relu_mask), ptr(ws),
                 ws_bytes, ptr(rows), ptr(b_k_chunks), ptr(c_tiles), int(getattr(c_tiles, '_gv_wanted', 0)) if c_tiles is not None else 0,
                 lib.stream())
        return out
    if live is not None and entry == 'gv_gemm_f32' and a.shape[0] == live[1] and a.device == live[0].device:
        lib.call('gv_gemm_f32_live_rows', 1 if trans_a else 0, 1 if trans_b else 0, m, n, k, ptr(a), lda, ptr(b), ldb, ptr(out),
                 out.stride(0) if m > 1 else n, ptr(bias), act, 1 if accumulate else 0, split_k, ptr(a_relu_mask), ptr(ws),
                 ws_bytes, ptr(live[0]), lib.stream())
        return out
    lib.call(entry, 1 if trans_a else 0, 1 if trans_b else 0, m, n, k, ptr(a), lda, ptr(b), ldb, ptr(out),
             out.stride(0) if m > 1 else n, ptr(bias), act, 1 if accumulate else 0, split_k, ptr(a_relu_mask), ptr(ws),
             ws_bytes, lib.stream())
    return out


def block_words(mask, kind):
    """Which blocks of a 0/1 matrix hold a non-zero entry, as the int64 words gv_gemm_f32_sparse takes (host-side: one read-back of the
    mask -- build them once per static mask).  ``mask`` (out, in), as a MaskedLinear stores it.
      'fwd'   y = x @ (W * mask)^T : per 64 outputs one word, bit c = inputs 16 c .. 16 c + 15 reach any of them
      'bwd'   g_x = g @ (W * mask) : per 64 inputs one word, bit c = outputs 16 c .. 16 c + 15 depend on any of them
      'tiles' dW = g^T @ x         : bit (i * ceil(in / 64) + j) = the 64 x 64 tile (i, j) of the (out, in) gradient is wanted"""
    import numpy as np
    m = (mask.detach().to('cpu').numpy() != 0)
    o, i = m.shape

    def pad(x, r, c):
        y = np.zeros((-(-x.shape[0] // r) * r, -(-x.shape[1] // c) * c), dtype=bool)
        y[:x.shape[0], :x.shape[1]] = x
        return y
    if kind in ('fwd', 'bwd'):
        mm = m if kind == 'fwd' else m.T                       # rows = the product's columns n, columns = its k
        if mm.shape[1] > 1024:
            return None
        y = pad(mm, 64, 16)
        blk = y.reshape(y.shape[0] // 64, 64, y.shape[1] // 16, 16).any(axis=(1, 3))       # (n tiles, k chunks)
        words = np.zeros(blk.shape[0], dtype=np.uint64)
        for c in range(blk.shape[1]):
            words |= blk[:, c].astype(np.uint64) << np.uint64(c)
    elif kind == 'tiles':
        y = pad(m, 64, 64)
        blk = y.reshape(y.shape[0] // 64, 64, y.shape[1] // 64, 64).any(axis=(1, 3)).reshape(-1)
        words = np.zeros((blk.size + 63) // 64, dtype=np.uint64)
        for b in np.nonzero(blk)[0]:
            words[b >> 6] |= np.uint64(1) << np.uint64(b & 63)
    else:
        raise ValueError(kind)
    out = torch.from_numpy(words.view(np.int64).copy()).to(mask.device)
    if kind == 'tiles':
        out._gv_wanted = int(blk.sum())          # how many tiles are wanted: lets the split-K product launch blocks for those only
    return out


def rank_scores(q, entities, target, bias=None):
    """Raw ranks (0-based, float: x.5 under ties) of ``target[i]`` among all entities under score = q @ entities^T + bias:
    the number of OTHER entities with a strictly larger logit plus HALF the number that tie with it (gv_rank_scores:
    MFMA tiles with a rank-count epilogue; the (m, V) score matrix is never stored).  Logits, not sigmoid outputs: same
    order, no saturation ties.  A NaN target score ranks last."""
    q, ld_q = _row_major(q, 'q')
    entities, ld_e = _row_major(entities, 'entities')
    if q.shape[1] != entities.shape[1]:
        raise ValueError('q / entities width mismatch')
    m, v = q.shape[0], entities.shape[0]
    target = target.reshape(-1)
    if target.numel() != m:
        raise ValueError('one target per query row')
    if m and (int(target.min()) < 0 or int(target.max()) >= v):
        raise ValueError(f'targets must lie in [0, {v})')
    tgt32 = target.to(device=q.device, dtype=torch.int32).contiguous()
    if bias is not None:
        bias = _chk(bias.reshape(1).to(torch.float32).contiguous(), name='bias')
    ws = torch.empty(max(m, 1), dtype=torch.float32, device=q.device)
    count = torch.empty(max(m, 1), dtype=torch.int32, device=q.device)
    lib.call('gv_rank_scores', ptr(q), ld_q, ptr(entities), ld_e, ptr(tgt32), ptr(bias), ptr(ws), ptr(count), m, v,
             q.shape[1], lib.stream())
    return count[:m].to(torch.float32) * 0.5


def pick_split_k(m_out, n_out, k):
    """Reduction-heavy shapes (weight gradients: small output, K = nodes) need split-K to fill 256 CUs."""
    tiles = ((m_out + 63) // 64) * ((n_out + 63) // 64)
    if tiles >= 256 or k < 2048:
        return 1
    if k >= 65536:      # a MADE's passes stacked: 500 x 500 over 131 k rows 676 us with 16 splits, 635 with 64 (1000 x 500: 1 348 -> 1 221)
        return 64
    return max(1, min(64, 1024 // tiles, k // 256))


def colsum(x, out=None, accumulate=False, relu_mask=None):
    x, ld = _row_major(x, 'x')
    if relu_mask is not None:
        relu_mask, ldm = _row_major(relu_mask, 'relu_mask')
        if tuple(relu_mask.shape) != tuple(x.shape) or ldm != ld:
            raise ValueError('relu_mask must have the shape and leading dimension of x')
    m, n = x.shape
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=x.device)
        accumulate = False
    ws = torch.empty(64 * n, dtype=torch.float32, device=x.device)
    lib.call('gv_colsum', ptr(x), ptr(relu_mask), m, n, ld, ptr(out), ptr(ws), 1 if accumulate else 0, lib.stream())
    return out


def epilogue_fwd(agg, addend, act, keep, keep_scale):
    agg = _chk(agg, name='agg')
    out = torch.empty_like(agg)
    if addend is not None:
        addend = _chk(addend, name='addend')
    lib.call('gv_rgcn_epilogue_fwd', ptr(agg), ptr(addend), act, ptr(keep), float(keep_scale), ptr(out),
             agg.shape[0], agg.shape[1], lib.stream())
    return out


def epilogue_bwd(out, grad_out, act, keep, keep_scale, colsum_out=None, colsum_accumulate=False):
    """g = d(act/dropout epilogue); with ``colsum_out`` the same pass also yields the column sums of g (bias grad)."""
    grad_out = _chk(grad_out.contiguous(), name='grad_out')
    g = torch.empty_like(grad_out)
    m, n = grad_out.shape
    fused = colsum_out is not None and n % 4 == 0 and n <= 1024
    part = torch.empty(EPILOGUE_COLSUM_SLICES * n, dtype=torch.float32, device=g.device) if fused else None
    lib.call('gv_rgcn_epilogue_bwd', ptr(out), ptr(grad_out), act, ptr(keep), float(keep_scale), ptr(g), m, n, ptr(part),
             lib.stream())
    if fused:
        lib.call('gv_colsum_finish', ptr(part), n, EPILOGUE_COLSUM_SLICES, ptr(colsum_out), 1 if colsum_accumulate else 0,
                 lib.stream())
    elif colsum_out is not None:
        colsum(g, out=colsum_out, accumulate=colsum_accumulate)
    return g


def axpby(alpha, x, beta=0.0, y=None, a=None):
    x = _chk(x.contiguous(), name='x')
    if y is None:
        y = torch.empty_like(x)
        beta = 0.0
    lib.call('gv_axpby', x.numel(), ptr(a), float(alpha), ptr(x), float(beta), ptr(y), lib.stream())
    return y


def copy_into(dst, src):
    """dst <- src (contiguous fp32, same size) as an ordinary KERNEL.  torch's copy_ / clone / cat of contiguous tensors go through
    hipMemcpyAsync, which a hipGraph records as a memcpy NODE -- and on this stack such a node can replay with stale parameters
    after any other runtime work between two replays ('Memory access fault by GPU'; DESIGN.md, hipGraph hazards).  Nothing on
    a capturable path may copy that way."""
    src = _chk(src.contiguous(), name='src')
    if dst.numel() != src.numel() or not dst.is_contiguous():
        raise ValueError('copy_into: same size, contiguous destination')
    lib.call('gv_axpby', src.numel(), None, 1.0, ptr(src), 0.0, ptr(dst), lib.stream())
    return dst


def copy_of(x):
    return copy_into(torch.empty_like(x, memory_format=torch.contiguous_format), x)


class _CatRows(torch.autograd.Function):
    """torch.cat([a, b], dim=0) of two (rows, d) matrices with kernel copies (see copy_into)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty(a.shape[0] + b.shape[0], a.shape[1], dtype=torch.float32, device=a.device)
        copy_into(out[:a.shape[0]], a)
        copy_into(out[a.shape[0]:], b)
        ctx.na = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.na], g[ctx.na:]


def cat_rows(a, b):
    return _CatRows.apply(a, b)


class _SplitRows(torch.autograd.Function):
    """(x[:n], x[n:]) whose backward assembles the gradient with kernel copies: autograd's own SliceBackward is zeros + copy_,
    i.e. a memcpy node in a captured step (see copy_into)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = int(n), tuple(x.shape)
        ctx.set_materialize_grads(False)
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        out = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        for part, g in ((out[:ctx.n], ga), (out[ctx.n:], gb)):
            if part.numel() == 0:
                continue
            if g is None:
                part.zero_()
            else:
                copy_into(part, g)
        return out, None


def split_rows(x, n):
    """The first ``n`` rows of ``x`` and the rest (differentiable; capturable)."""
    return _SplitRows.apply(x, n)


def mul(a, b):
    a, b = _chk(a.contiguous(), name='a'), _chk(b.contiguous(), name='b')
    out = torch.empty_like(a)
    lib.call('gv_mul', a.numel(), ptr(a), ptr(b), ptr(out), lib.stream())
    return out


def mul_multi(as_, bs, outs=None):
    """[a * b for a, b in zip(as_, bs)] in ONE launch (at most 8 pairs; gv_mul_multi).  ``outs``: per pair a contiguous
    tensor to write into, or None for a new one."""
    as_ = [_chk(a.contiguous(), name='a') for a in as_]
    bs = [_chk(b.contiguous(), name='b') for b in bs]
    outs = [torch.empty_like(a) if (outs is None or outs[i] is None) else outs[i] for i, a in enumerate(as_)]
    k = len(as_)
    tab = lambda ts: (_ct.c_void_p * k)(*[ptr(t) for t in ts])
    ta, tb, to = tab(as_), tab(bs), tab(outs)
    tn = (_ct.c_int64 * k)(*[a.numel() for a in as_])
    lib.call('gv_mul_multi', k, _ct.addressof(ta), _ct.addressof(tb), _ct.addressof(to), _ct.addressof(tn), lib.stream())
    return outs


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
# Device RNG: every random draw of a forward pass (dropout keep masks, reparameterisation noise, prior
# noise) comes from ONE launch of a counter-based generator (Philox4x32-10, gv_rng_fill) keyed by
# {seed, tick} held on the device.  The tick advances on the device (folded into the embedding lookup, or
# gv_rng_tick), so a captured hipGraph draws fresh numbers at every replay -- and none of torch's graph-safe
# RNG bookkeeping kernels (seed/offset fills, one generator launch per tensor) are on the path.
RNG_KEEP_MASK, RNG_NORMAL = 0, 1
_rng_streams = _it.count(1)
_rngs = {}


def new_rng_stream():
    """A process-unique stream id; modules take one per random draw site at construction."""
    return next(_rng_streams) & 0xFFFFFFFF


class DeviceRNG:
    def __init__(self, device):
        self.device = torch.device(device)
        self._seed = None
        self.state = torch.zeros(2, dtype=torch.int64, device=self.device)     # {seed, tick}
        self._used = set()          # streams drawn since the last tick
        self._sync_seed()

    def _sync_seed(self):
        seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF     # follows torch.manual_seed
        if seed != self._seed:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('torch.manual_seed changed during hipGraph capture')
            self._seed = seed
            self.state.copy_(torch.tensor([seed, 0], dtype=torch.int64))
            self._used.clear()

    def tick(self):
        self._sync_seed()
        lib.call('gv_rng_tick', ptr(self.state), lib.stream())
        self._used.clear()

    def folded_tick(self):
        """The state pointer for a kernel that advances the tick as a side effect (gv_gather_rows_rng_tick)."""
        self._sync_seed()
        self._used.clear()
        return self.state

    def fill(self, jobs):
        """jobs: [(tensor, kind, drop_p, stream)] -- uint8 tensors for RNG_KEEP_MASK, float32 for RNG_NORMAL."""
        self._sync_seed()
        if any(j[3] in self._used for j in jobs):     # same stream twice within one tick would repeat its numbers
            self.tick()
        for i in range(0, len(jobs), 8):
            part = jobs[i:i + 8]
            n = len(part)
            for t, kind, _, _ in part:
                want = torch.uint8 if kind == RNG_KEEP_MASK else torch.float32
                if t.dtype != want or not t.is_contiguous() or t.device != self.device:
                    raise TypeError('rng fill: buffers must be contiguous uint8 (mask) / float32 (normal) on the RNG device')
            ptrs = (_ct.c_void_p * n)(*[t.data_ptr() for t, _, _, _ in part])
            counts = (_ct.c_int64 * n)(*[t.numel() for t, _, _, _ in part])
            kinds = (_ct.c_int32 * n)(*[k for _, k, _, _ in part])
            ps = (_ct.c_float * n)(*[float(p) for _, _, p, _ in part])
            streams = (_ct.c_uint32 * n)(*[s for _, _, _, s in part])
            lib.call('gv_rng_fill', ptr(self.state), n, ptrs, counts, kinds, ps, streams, lib.stream())
        self._used.update(j[3] for j in jobs)


def device_rng(device):
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError('the device RNG lives on a ROCm device (no CPU fallback)')
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _rngs:
        _rngs[key] = DeviceRNG(torch.device('cuda', key))
    return _rngs[key]


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, tick_rng=None):
        table = _chk(table, name='embedding table')
        ids = _chk(ids.reshape(-1), torch.int64, 'node ids')
        out = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=table.device)
        if tick_rng is not None and ids.numel() > 0:      # the start-of-forward RNG tick rides on this launch
            lib.call('gv_gather_rows_rng_tick', ptr(table), ptr(ids), ptr(out), ids.numel(), table.shape[1],
                     ptr(tick_rng.folded_tick()), lib.stream())
        else:
            lib.call('gv_gather_rows', ptr(table), ptr(ids), ptr(out), ids.numel(), table.shape[1], lib.stream())
        ctx.save_for_backward(ids)
        ctx.shape = tuple(table.shape)
        ctx.direct = _direct(table)
        _stamp_direct(ctx)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = _chk(g.contiguous(), name='grad')
        _verify_direct(ctx)
        tgt = ctx.direct
        gt = tgt if tgt is not None else torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        lib.call('gv_scatter_add_rows', ptr(g), ptr(ids), ptr(gt), ids.numel(), ctx.shape[1], lib.stream())
        if tgt is not None:
            GRAD_FRESH.discard(tgt.data_ptr())
        return (None if tgt is not None else gt), None, None


class _EmbeddingIdentity(torch.autograd.Function):
    """The lookup when ids == arange(num_rows) (full-graph training, kgvae/link_predict.py:141-147): no gathered copy --
    the output aliases the table -- and no scatter-add in backward: the consuming layer writes the gradient rows straight
    into the table's gradient (``_gv_grad_target`` on the output tells it where), or, failing that, one add."""

    @staticmethod
    def forward(ctx, table, tick_rng):
        ctx.set_materialize_grads(False)
        if tick_rng is not None:
            tick_rng.tick()
        ctx.direct = _direct(table)
        _stamp_direct(ctx)
        return table.detach().view(table.shape)

    @staticmethod
    def backward(ctx, g):
        if g is None:                  # the layer already wrote the rows into the gradient arena
            return None, None
        g = _chk(g.contiguous(), name='grad')
        _verify_direct(ctx)
        tgt = ctx.direct
        if tgt is None:
            return g, None
        lib.call('gv_axpby', g.numel(), None, 1.0, ptr(g), 1.0, ptr(tgt), lib.stream())
        GRAD_FRESH.discard(tgt.data_ptr())
        return None, None


_identity_ids = {}


def _ids_are_identity(table, ids):
    """ids == arange(table rows)?  Decided once per ids tensor (one host synchronisation), cached on identity + version."""
    if ids.numel() != table.shape[0]:
        return False
    key = (ids.data_ptr(), ids._version, ids.numel())
    hit = _identity_ids.get(key)
    if hit is None:
        if len(_identity_ids) > 64:
            _identity_ids.clear()
        # the entry holds the ids tensor: its address cannot be handed to another tensor while the verdict is cached
        hit = _identity_ids[key] = (bool((ids.reshape(-1) == torch.arange(ids.numel(), device=ids.device)).all()), ids)
    return hit[0]


def embedding(table, ids, tick_rng=None, sole_consumer=False):
    """table[ids].  ``sole_consumer``: the caller guarantees that ONE R-GCN layer consumes the result (the encoders' input
    layer): with an identity lookup that layer may then write dL/dx straight into the table's gradient rows."""
    identity = False
    if table.is_cuda and isinstance(ids, torch.Tensor) and ids.is_cuda and ids.dtype == torch.int64:
        if torch.cuda.is_current_stream_capturing():      # no host synchronisation under capture: only a cached verdict counts
            hit = _identity_ids.get((ids.data_ptr(), ids._version, ids.numel()))
            identity = bool(hit and hit[0]) and ids.numel() == table.shape[0]
        else:
            identity = _ids_are_identity(table, ids)
    if identity and sole_consumer:      # the output ALIASES the table: only for the fused layer, which never writes its input
        out = _EmbeddingIdentity.apply(table, tick_rng)
        out._gv_grad_target = _direct(table)
        return out
    return _Embedding.apply(table, ids, tick_rng)      # a real copy, as nn.Embedding returns


class _RelGraphConvBdd(torch.autograd.Function):
    """out = keep*scale*act( sum_e norm_e * blockdiag(W_{r_e}) x[src_e]  + x@loop_weight + h_bias ).

    ``reduce_hook`` (multi-GPU edge sharding): a callable that STARTS an in-place sum of a tensor over the
    ranks and returns a handle with ``wait()`` (distributed.make_reduce_hook).  With a hook the raw aggregate
    is kept separate from the epilogue so that it can be all-reduced in between, overlapped with the layer's
    self-loop GEMM; backward all-reduces the gradient of the aggregate the same way, overlapped with the bias /
    loop-weight gradients and the loop GEMM.
    """

    @staticmethod
    def forward(ctx, x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, act, keep, keep_scale,
                reduce_hook):
        x, _ = _row_major(x, 'x')
        n, in_feat = x.shape
        si = in_feat // num_bases
        so = weight.shape[1] // (num_bases * si)
        out_feat = num_bases * so
        coef = None if norm is None else norm.reshape(-1)
        # relation phases (weights staged through LDS once per tile, csrc/k_phase.hip): static single-GPU graphs
        tl = None
        if reduce_hook is None and use_phases(gidx, si, so, False, x.shape[0], in_feat):
            tl = ridx.phase_order(gidx, 'dst', num_bases, si, so)
        ctx.tiles = tl is not None
        # lane-packed weights pay off once a block's weights span >= 32 B (measured: 2x4 / 4x2 blocks -24 % / -19 %,
        # 2x2 blocks +-0): pack per launch kind, a ~1.5 MB pass per layer
        pk = not ctx.tiles and si * so >= 8 and pack_supported(num_bases, si, so, False)
        bwd_phases = reduce_hook is None and use_phases(gidx, so, si, True, n, out_feat)
        pk_bwd = not bwd_phases and si * so >= 8 and pack_supported(num_bases, so, si, True)
        ctx.w_bwd_packed = None
        if pk and pk_bwd and ctx.needs_input_grad[0]:      # one launch writes both layouts; backward-x reuses its copy
            w_fwd, ctx.w_bwd_packed = torch.empty_like(weight), torch.empty_like(weight)
            lib.call('gv_rgcn_bdd_pack_weight_pair', ptr(weight), weight.shape[0], num_bases, si, so, ptr(w_fwd),
                     ptr(ctx.w_bwd_packed), lib.stream())
        else:
            w_fwd = pack_weight(weight, num_bases, si, so, False) if pk else weight

        ctx.loop_bf = dense_bf16_forms(loop_weight) if (loop_weight is not None and x.shape[0] >= 4096) else None

        def self_loop_term():
            if loop_weight is not None and ctx.loop_bf is not None:
                return dense_bf16(x, ctx.loop_bf[0], loop_weight.shape[1], loop_weight.shape[0], bias=h_bias)
            if loop_weight is not None:
                return gemm(x, loop_weight, bias=h_bias)
            if h_bias is not None:
                return h_bias.unsqueeze(0).expand(n, out_feat).contiguous()
            return None

        ctx.grouped = not ctx.tiles and reduce_hook is None and use_relation_groups(weight, gidx)
        lp = lds_plan(weight.shape[0], num_bases, si, so) if (not ctx.tiles and not ctx.grouped and reduce_hook is None) else None
        if lp is not None and not lds_graph(gidx, 'dst', lp[2]):
            lp = None
        ctx.k1_bf = bool(lp[3]) if lp is not None else False
        if lp is not None and not torch.cuda.is_current_stream_capturing():
            # the backward-x launch's list too, while a host read-back is still allowed: a step captured after this eager forward
            # (a no-grad evaluation pass in front of the training capture) must not find it missing
            lp_b = lds_plan(weight.shape[0], num_bases, so, si, bf=ctx.k1_bf)
            if lp_b is not None:
                gidx.lds_order('src', lp_b[2])
        if lp is not None:       # few relation types: the whole table resident in LDS (csrc/k_lds.hip)
            out = bdd_aggregate_lds(gidx.lds_order('dst', lp[2]), gidx.nbr_by_dst, ridx.et_by_dst, coef, gidx.by_dst.perm, x,
                                    pack_weight_lds(weight, num_bases, si, so, False, lp), weight.shape[0], num_bases, si, so,
                                    False, self_loop_term(), act, keep, keep_scale, plan=lp)
        elif ctx.tiles:
            out = bdd_aggregate_phases(tl, None if coef is None else tl.coef(coef), x,
                                       pack_weight_phase(tl, weight, num_bases, si, so), weight.shape[0], num_bases, si, so,
                                       self_loop_term(), act, keep, keep_scale)
        elif ctx.grouped:
            seg, nbr, ety, perm = ridx.grouped_order(gidx, 'dst')
            coef_g = None if coef is None else ridx.grouped_coef(coef, 'dst', perm)
            out = bdd_aggregate(seg, nbr, ety, coef_g, None, x, w_fwd, num_bases, si, so, False, self_loop_term(), act, keep,
                                keep_scale, packed=pk)
        elif reduce_hook is None and K1_REL_RUNS and not gidx.sync_free:
            nbr_r, et_r, eid_r, coef_r = ridx.rel_sorted(gidx, 'dst', coef)      # rows sorted by relation: weight reuse along runs
            out = bdd_aggregate(_k1_items(gidx, gidx.by_dst.seg), nbr_r, et_r, coef_r, None, x, w_fwd,
                                num_bases, si, so, False, self_loop_term(), act, keep, keep_scale, packed=pk)
        elif reduce_hook is None:
            out = bdd_aggregate(_k1_items(gidx, gidx.by_dst.seg), gidx.nbr_by_dst, ridx.et_by_dst, coef, gidx.by_dst.perm, x, w_fwd,
                                num_bases, si, so, False, self_loop_term(), act, keep, keep_scale, packed=pk)
        else:
            # edge-sharded: aggregate the destination rows in DIST_FWD_CHUNKS blocks; the all-reduce of block c runs
            # on RCCL's stream while block c+1 is aggregated and, at the end, under the self-loop GEMM
            agg = torch.empty(n, out_feat, dtype=torch.float32, device=x.device)
            pending = []
            for r0, r1, sub in gidx.dst_chunks(DIST_FWD_CHUNKS):
                bdd_aggregate(sub, gidx.nbr_by_dst, ridx.et_by_dst, coef, gidx.by_dst.perm, x, w_fwd, num_bases, si, so,
                              packed=pk, out=agg)
                pending.append(reduce_hook(agg[r0:r1]))
            addend = self_loop_term()
            for h in pending:
                h.wait()
            out = epilogue_fwd(agg, addend, act, keep, keep_scale)
        ctx.save_for_backward(x, weight, loop_weight, coef, out if act == ACT_RELU else None, keep)
        ctx.meta = (gidx, ridx, num_bases, si, so, act, keep_scale, h_bias is not None, reduce_hook)
        ctx.w_version = weight._version
        ctx.direct = (_direct(weight), _direct(h_bias), _direct(loop_weight))
        _stamp_direct(ctx)
        # x is the embedding table itself (identity lookup) and this layer is its only consumer: dL/dx rows go straight
        # into the table's gradient (zero at this point of the step: the optimiser cleared it, nothing else adds to it)
        tgt = getattr(x, '_gv_grad_target', None)
        ctx.x_grad_target = tgt if (tgt is not None and reduce_hook is None and tuple(tgt.shape) == tuple(x.shape)) else None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, weight, loop_weight, coef, out, keep = ctx.saved_tensors
        gidx, ridx, nb, si, so, act, keep_scale, has_bias, reduce_hook = ctx.meta
        _verify_direct(ctx)
        d_w, d_b, d_l = ctx.direct
        grad_bias = None
        if has_bias and ctx.needs_input_grad[2]:         # bias gradient = column sums of g, from the same pass
            grad_bias = d_b if d_b is not None else torch.empty(grad_out.shape[1], dtype=torch.float32, device=x.device)
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale, colsum_out=grad_bias, colsum_accumulate=d_b is not None)
            if d_b is not None:
                grad_bias = None
        else:
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale)
        pending = None
        g_agg = g
        if reduce_hook is not None:      # gradient of this rank's partial aggregate = sum over ranks; overlapped below
            g_agg = copy_of(g)
            pending = reduce_hook(g_agg)
        grad_loop = gx_loop = None
        # the layer's two weight gradients (stored straight into the optimiser's arena: nothing in this backward pass reads them)
        # on the side stream, beside the dL/dx path -- where that pays (RGCN_BWD_SIDE)
        side_w = reduce_hook is None and d_l is not None and d_w is not None and loop_weight is not None \
            and ctx.needs_input_grad[3] and ctx.needs_input_grad[1] and rgcn_bwd_side(gidx.num_edges, max(x.shape[1], g.shape[1]))
        if side_w:
            with backward_side(True, x, g, g_agg, weight, coef, rgcn=True):
                gw_first = rgcn_side_gradw_first(gidx.num_edges, x.shape[1])
                if not gw_first:
                    gemm(x, g, trans_a=True, split_k=pick_split_k(x.shape[1], g.shape[1], x.shape[0]), out=d_l, accumulate=True)
                static = not gidx.sync_free and coef is not None
                coef_r, idx_r = (ridx.coef_in_rel_order(coef), None) if static else (coef, ridx.by_rel.perm)
                bdd_grad_weight(ridx.by_rel.seg, ridx.src_by_rel, ridx.dst_by_rel, coef_r, idx_r, x, g_agg, nb, si, so, out=d_w, accumulate=True)
                if gw_first:
                    gemm(x, g, trans_a=True, split_k=pick_split_k(x.shape[1], g.shape[1], x.shape[0]), out=d_l, accumulate=True)
        if loop_weight is not None:
            if ctx.needs_input_grad[3] and not side_w:
                # (on this stream, NOT ordered behind the side stream: d_l / d_w are this layer's own arena slices, which no
                # other node of the pass writes -- a wait here serialises the flows' weight-gradient products with the R-GCN
                # backward: WN18RR + 3 IAF 4.9 -> 5.2 ms when it was tried)
                grad_loop = gemm(x, g, trans_a=True, split_k=pick_split_k(x.shape[1], g.shape[1], x.shape[0]),
                                 out=d_l, accumulate=d_l is not None)
                if d_l is not None:
                    grad_loop = None
            if ctx.needs_input_grad[0]:
                if getattr(ctx, 'loop_bf', None) is not None:
                    gx_loop = dense_bf16(g, ctx.loop_bf[1], loop_weight.shape[0], loop_weight.shape[1])
                else:
                    gx_loop = gemm(g, loop_weight, trans_b=True)
        if pending is not None:
            pending.wait()
        grad_x = None
        tl = None
        if reduce_hook is None and ctx.needs_input_grad[0] and use_phases(gidx, so, si, True, g_agg.shape[0], g_agg.shape[1]):
            tl = ridx.phase_order(gidx, 'src', nb, so, si)
        # dL/dx rows may be stored straight into the embedding table's gradient (identity lookup) -- but only while that
        # buffer is known to be zero (fresh from zero_grad / step); a second backward before the next step, or another
        # lookup of the table whose gradient landed first, must be added to, not erased
        x_tgt = x_add = ctx.x_grad_target if ctx.needs_input_grad[0] else None
        if x_tgt is not None:
            if x_tgt.data_ptr() in GRAD_FRESH:
                GRAD_FRESH.discard(x_tgt.data_ptr())
                x_add = None
            else:
                x_tgt = None
        if tl is not None:
            grad_x = bdd_aggregate_phases(tl, None if coef is None else tl.coef(coef), g_agg,
                                          pack_weight_phase(tl, weight, nb, so, si), weight.shape[0], nb, so, si, gx_loop,
                                          out=x_tgt)
        elif ctx.needs_input_grad[0] and not ctx.grouped and reduce_hook is None and \
                lds_plan(weight.shape[0], nb, so, si, bf=ctx.k1_bf) is not None and \
                lds_graph(gidx, 'src', lds_plan(weight.shape[0], nb, so, si, bf=ctx.k1_bf)[2]):
            lp = lds_plan(weight.shape[0], nb, so, si, bf=ctx.k1_bf)
            static = not gidx.sync_free and coef is not None
            coef_s, idx_s = (gidx.coef_in_src_order(coef), None) if static else (coef, gidx.by_src.perm)
            grad_x = bdd_aggregate_lds(gidx.lds_order('src', lp[2]), gidx.nbr_by_src, ridx.et_by_src, coef_s, idx_s, g_agg,
                                       pack_weight_lds(weight, nb, so, si, True, lp), weight.shape[0], nb, so, si, True,
                                       gx_loop, out=x_tgt, plan=lp)
        elif ctx.needs_input_grad[0]:
            pk = si * so >= 8 and pack_supported(nb, so, si, True)
            if ctx.w_bwd_packed is not None and weight._version == ctx.w_version:
                w_bwd = ctx.w_bwd_packed
            else:
                w_bwd = pack_weight(weight, nb, so, si, True) if pk else weight
            # static graphs: the edge norm is cached in this launch's order; per-batch graphs read it through the index
            static = not gidx.sync_free and coef is not None
            if ctx.grouped:
                seg, nbr, ety, perm = ridx.grouped_order(gidx, 'src')
                coef_g = None if coef is None else ridx.grouped_coef_src(coef, perm)
                grad_x = bdd_aggregate(seg, nbr, ety, coef_g, None, g_agg, w_bwd, nb, so, si, True, gx_loop, out=x_tgt, packed=pk)
            elif K1_REL_RUNS and not gidx.sync_free and reduce_hook is None:
                nbr_r, et_r, eid_r, coef_r = ridx.rel_sorted(gidx, 'src', coef)
                grad_x = bdd_aggregate(_k1_items(gidx, gidx.by_src.seg), nbr_r, et_r, coef_r, None, g_agg,
                                       w_bwd, nb, so, si, True, gx_loop, out=x_tgt, packed=pk)
            else:
                coef_s, idx_s = (gidx.coef_in_src_order(coef), None) if static else (coef, gidx.by_src.perm)
                grad_x = bdd_aggregate(_k1_items(gidx, gidx.by_src.seg), gidx.nbr_by_src, ridx.et_by_src, coef_s, idx_s, g_agg,
                                       w_bwd, nb, so, si, True, gx_loop, out=x_tgt, packed=pk)
        if grad_x is not None and x_tgt is not None:
            grad_x = None                                  # written in place
        elif grad_x is not None and x_add is not None:     # the table's gradient already holds something: add
            lib.call('gv_axpby', grad_x.numel(), None, 1.0, ptr(grad_x), 1.0, ptr(x_add), lib.stream())
            grad_x = None
        grad_w = None
        if ctx.needs_input_grad[1] and not side_w:
            static = not gidx.sync_free and coef is not None
            coef_r, idx_r = (ridx.coef_in_rel_order(coef), None) if static else (coef, ridx.by_rel.perm)
            grad_w = bdd_grad_weight(ridx.by_rel.seg, ridx.src_by_rel, ridx.dst_by_rel, coef_r, idx_r, x,
                                     g_agg, nb, si, so, out=d_w, accumulate=d_w is not None)
            if d_w is not None:
                grad_w = None
        return grad_x, grad_w, grad_bias, grad_loop, None, None, None, None, None, None, None, None


K1_REL_RUNS = _os.environ.get('GV_K1_REL_RUNS', '1') == '1'     # static graphs: rows sorted by relation (weight reuse along runs)
REL_GROUPS = _os.environ.get('GV_REL_GROUPS', '0')         # '0' (default: off) | 'auto' | '1'
L2_WEIGHT_BUDGET = 3 << 20                                  # bytes of relation weights one XCD's 4 MiB L2 can keep hot


def use_relation_groups(weight, gidx):
    """Group a row's edges by relation range (RelationIndex.grouped_order) when the weight table overflows an XCD's L2:
    static graphs only (the grouped index is built with host-side sizes).  Opt-in: measured at h = 500 it takes 19 % off
    the 5x10 aggregation launch but is step-neutral, because every row then goes through the fix-up pass."""
    if REL_GROUPS == '0' or gidx.sync_free or gidx.num_edges == 0:
        return False
    return REL_GROUPS == '1' or weight.numel() * 4 > L2_WEIGHT_BUDGET


def rel_graph_conv_bdd(x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, act=ACT_NONE, keep=None,
                       keep_scale=1.0, reduce_hook=None):
    return _RelGraphConvBdd.apply(x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, act, keep,
                                  float(keep_scale), reduce_hook)


def rel_rows_gemm(feat, rows, w3, transpose_w, tiles, n_tiles, n_edges):
    """msg[p] = feat[rows[p]] @ W_r (or W_r^T) for the edges in by-relation order; w3 is (R, in, out)."""
    feat, ld = _row_major(feat, 'feat')
    w3 = _chk(w3, name='dense relation weights')
    r, fin, fout = w3.shape
    msg = torch.empty(n_edges, fin if transpose_w else fout, dtype=torch.float32, device=feat.device)
    lib.call('gv_rel_rows_gemm', ptr(feat), ld, ptr(rows), ptr(w3), r, fin, fout, 1 if transpose_w else 0, ptr(tiles), n_tiles,
             ptr(msg), lib.stream())
    return msg


class _RelGraphConvDense(torch.autograd.Function):
    """RelGraphConv with a full (in x out) matrix per relation -- the `basis` regulariser after W_r = sum_b w_comp[r,b] V_b
    (DGL's basis_message_func).  Messages are relation-grouped f32 MFMA GEMMs with gathered rows
    (gv_rel_rows_gemm, edges in by-relation order); their per-destination sum, x norm, + self loop + bias, activation and
    dropout is the K1 aggregation with 1x1 blocks over the message rows.  Backward: the same two steps transposed for
    dL/dx, one relation-grouped (in x E_r) @ (E_r x out) product per relation for dL/dW."""

    @staticmethod
    def forward(ctx, x, w3, h_bias, loop_weight, norm, gidx, ridx, act, keep, keep_scale):
        x, _ = _row_major(x, 'x')
        n, fin = x.shape
        r, fin_w, fout = w3.shape
        if fin_w != fin:
            raise ValueError(f'dense relation weights are {tuple(w3.shape)}, x has {fin} columns')
        coef = None if norm is None else norm.reshape(-1)
        tiles, n_tiles, pos_d, pos_s, zeros = ridx.dense_plan(gidx)
        addend = None
        if loop_weight is not None:
            addend = gemm(x, loop_weight, bias=h_bias)
        elif h_bias is not None:
            addend = h_bias.unsqueeze(0).expand(n, fout).contiguous()
        msg = rel_rows_gemm(x, ridx.src_by_rel, w3, False, tiles, n_tiles, gidx.num_edges)
        ones = torch.ones(1, fout, dtype=torch.float32, device=x.device)
        out = bdd_aggregate(gidx.by_dst.seg, pos_d, zeros, coef, gidx.by_dst.perm, msg, ones, fout, 1, 1, False, addend, act,
                            keep, keep_scale)
        ctx.save_for_backward(x, w3, loop_weight, coef, out if act == ACT_RELU else None, keep)
        ctx.meta = (gidx, ridx, act, keep_scale, h_bias is not None)
        ctx.direct = (_direct(h_bias), _direct(loop_weight))
        _stamp_direct(ctx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, w3, loop_weight, coef, out, keep = ctx.saved_tensors
        gidx, ridx, act, keep_scale, has_bias = ctx.meta
        _verify_direct(ctx)
        d_b, d_l = ctx.direct
        r, fin, fout = w3.shape
        tiles, n_tiles, pos_d, pos_s, zeros = ridx.dense_plan(gidx)
        grad_bias = grad_loop = grad_x = grad_w = None
        if has_bias and ctx.needs_input_grad[2]:
            grad_bias = d_b if d_b is not None else torch.empty(fout, dtype=torch.float32, device=x.device)
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale, colsum_out=grad_bias, colsum_accumulate=d_b is not None)
            if d_b is not None:
                grad_bias = None
        else:
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale)
        gx_loop = None
        if loop_weight is not None:
            if ctx.needs_input_grad[3]:
                grad_loop = gemm(x, g, trans_a=True, split_k=pick_split_k(fin, fout, x.shape[0]), out=d_l,
                                 accumulate=d_l is not None)
                if d_l is not None:
                    grad_loop = None
            if ctx.needs_input_grad[0]:
                gx_loop = gemm(g, loop_weight, trans_b=True)
        if ctx.needs_input_grad[0]:
            msg2 = rel_rows_gemm(g, ridx.dst_by_rel, w3, True, tiles, n_tiles, gidx.num_edges)
            ones = torch.ones(1, fin, dtype=torch.float32, device=x.device)
            coef_s, idx_s = (gidx.coef_in_src_order(coef), None) if coef is not None else (None, None)
            grad_x = bdd_aggregate(gidx.by_src.seg, pos_s, zeros, coef_s, idx_s, msg2, ones, fin, 1, 1, False, gx_loop)
        if ctx.needs_input_grad[1]:
            g2, ld_g = _row_major(g, 'g')
            grad_w = torch.empty_like(w3)
            scale = ridx.coef_in_rel_order(coef) if coef is not None else None
            lib.call('gv_rel_gradw_gemm', ptr(x), x.stride(0), ptr(ridx.src_by_rel), ptr(g2), ld_g, ptr(ridx.dst_by_rel),
                     ptr(scale), ptr(ridx.by_rel.seg.rowptr), r, fin, fout, ptr(grad_w), lib.stream())
        return grad_x, grad_w, grad_bias, grad_loop, None, None, None, None, None, None


class _RelGraphConvSelect(torch.autograd.Function):
    """RelGraphConv('basis') on INTEGER-ID node features (the one-hot input layer of kgvae/entity_classify.py:25-34, :63;
    DGL utils.bmm_maybe_select): a message is ROW (etype, id[src]) of the relation's (in x out) matrix, no product --

        out[v] = keep*scale*act( sum_{e: dst=v} norm_e * W[etype_e * in + id[src_e], :]  +  loop_rows[v] + h_bias )

    Forward = the K1 aggregation with 1x1 blocks gathering rows of W (R*in, out) through a per-edge row index; backward
    w.r.t. W = the same aggregation over the transposed incidence (destination = W row, source = node row of dL/dh), a
    rectangular graph index built once per (graph, ids); ``loop_rows`` = loop_weight[id] comes in through ops.embedding,
    whose backward scatter-adds its gradient."""

    @staticmethod
    def forward(ctx, wflat, loop_rows, h_bias, norm, gidx, plan, act, keep, keep_scale):
        wflat, _ = _row_major(wflat, 'relation rows')
        n, fout = gidx.num_nodes, wflat.shape[1]
        coef = None if norm is None else norm.reshape(-1)
        addend = None
        if h_bias is not None:
            addend = h_bias.unsqueeze(0).expand(n, fout).contiguous()
            if loop_rows is not None:
                lib.call('gv_axpby', addend.numel(), None, 1.0, ptr(_chk(loop_rows.contiguous(), name='loop rows')), 1.0,
                         ptr(addend), lib.stream())
        elif loop_rows is not None:
            addend = _chk(loop_rows.contiguous(), name='loop rows')
        ones = torch.ones(1, fout, dtype=torch.float32, device=wflat.device)
        out = bdd_aggregate(gidx.by_dst.seg, plan['row_by_dst'], plan['zeros'], coef, gidx.by_dst.perm, wflat, ones, fout, 1, 1,
                            False, addend, act, keep, keep_scale)
        ctx.save_for_backward(coef, out if act == ACT_RELU else None, keep)
        ctx.meta = (plan, act, keep_scale, h_bias is not None, loop_rows is not None, tuple(wflat.shape))
        ctx.direct_b = _direct(h_bias)
        _stamp_direct(ctx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coef, out, keep = ctx.saved_tensors
        plan, act, keep_scale, has_bias, has_loop, wshape = ctx.meta
        _verify_direct(ctx)
        d_b = ctx.direct_b
        grad_bias = None
        if has_bias and ctx.needs_input_grad[2]:
            grad_bias = d_b if d_b is not None else torch.empty(grad_out.shape[1], dtype=torch.float32, device=grad_out.device)
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale, colsum_out=grad_bias, colsum_accumulate=d_b is not None)
            if d_b is not None:
                grad_bias = None
        else:
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale)
        grad_w = None
        if ctx.needs_input_grad[0]:
            gi = plan['by_row']                      # destinations = rows of W, sources = node rows of g
            ones = torch.ones(1, wshape[1], dtype=torch.float32, device=g.device)
            coef_r = None if coef is None else coef[plan['edge_of_by_row']].contiguous()
            grad_w = bdd_aggregate(gi.by_dst.seg, gi.nbr_by_dst, plan['zeros'], coef_r, None, g, ones, wshape[1], 1, 1)
        return grad_w, (g if has_loop else None), grad_bias, None, None, None, None, None, None


def select_plan(gidx, ridx, ids, in_feat):
    """Index of the row-select layer, once per (graph, relation types, ids): the W row of every edge in by-destination
    order, and the rectangular graph (W row <- destination node) of its weight gradient."""
    cache = ridx.__dict__.setdefault('_select', {})
    key = (ids.data_ptr(), ids._version, ids.numel(), int(in_feat))
    hit = cache.get(key)
    if hit is None:
        ids = ids.reshape(-1)
        if ids.numel() != gidx.num_src_nodes:
            raise ValueError(f'{ids.numel()} node ids for a graph of {gidx.num_src_nodes} nodes')
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= in_feat):
            raise ValueError(f'integer node features must lie in [0, in_feat = {in_feat})')
        et = ridx.keepalive.reshape(-1).to(torch.int64)
        row = et * in_feat + ids[gidx.src32.long()]                         # caller's edge order
        if int(ridx.num_rels) * int(in_feat) >= 2 ** 31:
            raise ValueError('num_rels * in_feat must fit int32')
        perm_d = gidx.by_dst.perm
        row_by_dst = (row if perm_d is None else row[perm_d.long()]).to(torch.int32).contiguous()
        # weight gradient: a graph whose destinations are W rows and whose sources are the edges' destination NODES
        by_row = GraphIndex(gidx.dst32.long(), row, int(ridx.num_rels) * int(in_feat), num_src_nodes=gidx.num_nodes)
        edge_of = torch.arange(gidx.num_edges, device=gidx.device) if by_row.by_dst.perm is None else by_row.by_dst.perm.long()
        hit = cache[key] = dict(row_by_dst=row_by_dst, by_row=by_row, edge_of_by_row=edge_of,
                                zeros=torch.zeros(max(gidx.num_edges, 1), dtype=torch.int32, device=gidx.device), keep=ids)
    return hit


def rel_graph_conv_select(ids, w3, h_bias, loop_weight, norm, gidx, ridx, act=ACT_NONE, keep=None, keep_scale=1.0):
    """RelGraphConv('basis') with integer-id features: w3 is (R, in, out), ids int64 (N,)."""
    r, fin, fout = w3.shape
    plan = select_plan(gidx, ridx, ids, fin)
    loop_rows = embedding(loop_weight, ids.reshape(-1)) if loop_weight is not None else None
    return _RelGraphConvSelect.apply(w3.reshape(r * fin, fout), loop_rows, h_bias, norm, gidx, plan, act, keep, float(keep_scale))


def rel_graph_conv_dense(x, w3, h_bias, loop_weight, norm, gidx, ridx, act=ACT_NONE, keep=None, keep_scale=1.0):
    return _RelGraphConvDense.apply(x, w3, h_bias, loop_weight, norm, gidx, ridx, act, keep, float(keep_scale))


class _RelGraphConvRows(torch.autograd.Function):
    """The same layer for the multi-GPU DESTINATION-ROW partition ("cheaper alternative"): this rank owns
    ``part.own_rows`` rows of the node table and every edge that ends in them, so it computes FINAL output rows -- no
    reduction of partial aggregates.  ``gidx`` is rectangular: destinations are local row ids, sources index the whole
    table (``part.total_rows`` rows = world x slot rows; a rank's real rows come first in its slot, the tail is zero).

      gather_input   x is this rank's slot (slot_rows, in): ALL-GATHER it into the full table, under the self-loop
                     GEMM of the rank's own rows and the weight packing; backward REDUCE-SCATTERs the partial gradient
                     of the full table (K1^T over the rank's edges), under the loop-weight products and grad-W
      otherwise      x already is the full table (replicated embedding lookup); backward returns this rank's partial
                     gradient of all its rows (summed later by the parameter-gradient all-reduce)
      pad_output     return a (slot_rows, out) tensor with a zero tail, ready to be gathered by the next layer
      gather_output  PIPELINED exchange (part.chunks > 1): the layer itself gathers its output -- its rows are aggregated in
                     part.chunks blocks of slot rows and block k's all-gather runs under block k + 1's aggregation -- and returns
                     the FULL table (total_rows, out).  The gradient that comes back for it is the consumer's reduce-scattered
                     partial sums, valid in this rank's slot rows only (see x_gathered)
      x_gathered     x is such a gathered table: no gather here; backward computes K1^T block by block (the same slot-row range
                     of every rank per block), reduce-scatters block k under block k + 1 and returns a full-size tensor whose
                     rows of THIS rank's slot hold the summed gradient (the others are never read)
    """

    @staticmethod
    def forward(ctx, x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, act, keep, keep_scale, part,
                gather_input, pad_output, gather_output=False, x_gathered=False):
        x, _ = _row_major(x, 'x')
        c, slot, row0, total = part.own_rows, part.slot_rows, part.row0, part.total_rows
        in_feat = x.shape[1]
        si = in_feat // num_bases
        so = weight.shape[1] // (num_bases * si)
        out_feat = num_bases * so
        if gidx.num_nodes != c or gidx.num_src_nodes != total:
            raise ValueError(f'row-partition graph index is {gidx.num_nodes} x {gidx.num_src_nodes}, expected {c} x {total}')
        if x_gathered and gather_input:
            raise ValueError('x_gathered: the table arrives gathered, gather_input must be False')
        if x.shape[0] != (slot if gather_input else total):
            raise ValueError(f'x has {x.shape[0]} rows, expected {slot if gather_input else total}')
        coef = None if norm is None else norm.reshape(-1)
        pending = None
        if gather_input:
            x_full = torch.empty(total, in_feat, dtype=torch.float32, device=x.device)
            pending = part.all_gather(x_full, x)
            x_own = x[:c]
        else:
            x_full, x_own = x, x[row0:row0 + c]
        pk = si * so >= 8 and pack_supported(num_bases, si, so, False)
        pk_bwd = si * so >= 8 and pack_supported(num_bases, so, si, True)
        ctx.w_bwd_packed = None
        if pk and pk_bwd:
            w_fwd, ctx.w_bwd_packed = torch.empty_like(weight), torch.empty_like(weight)
            lib.call('gv_rgcn_bdd_pack_weight_pair', ptr(weight), weight.shape[0], num_bases, si, so, ptr(w_fwd),
                     ptr(ctx.w_bwd_packed), lib.stream())
        else:
            w_fwd = pack_weight(weight, num_bases, si, so, False) if pk else weight
        addend = None
        if c > 0:
            if loop_weight is not None:
                addend = gemm(x_own, loop_weight, bias=h_bias)
            elif h_bias is not None:
                addend = h_bias.unsqueeze(0).expand(c, out_feat).contiguous()
        pad_output = pad_output or gather_output
        buf = torch.empty(slot if pad_output else c, out_feat, dtype=torch.float32, device=x.device)
        if pad_output and c < slot:
            buf[c:].zero_()
        if pending is not None:
            pending.wait()
        out = buf[:c]
        result = buf
        if gather_output:
            # block k of the rank's rows is final (aggregate + epilogue in one launch) -> its gather starts; block k + 1 follows
            result = torch.empty(total, out_feat, dtype=torch.float32, device=x.device)
            bounds = part.chunk_bounds()
            subs = gidx.row_blocks('dst', [[(min(a, c), min(b, c))] for a, b in bounds])
            waits = []
            for (a, b), sub in zip(bounds, subs):
                if c > 0 and sub.n_items > 0:
                    bdd_aggregate(sub, gidx.nbr_by_dst, ridx.et_by_dst, coef, gidx.by_dst.perm, x_full, w_fwd, num_bases,
                                  si, so, False, addend, act, keep, keep_scale, out=out, packed=pk)
                waits.append(part.gather_rows(result, buf, a, b))
            for wt in waits:
                wt.wait()
        elif c > 0:
            bdd_aggregate(gidx.by_dst.seg, gidx.nbr_by_dst, ridx.et_by_dst, coef, gidx.by_dst.perm, x_full, w_fwd, num_bases,
                          si, so, False, addend, act, keep, keep_scale, out=out, packed=pk)
        ctx.save_for_backward(x_full, weight, loop_weight, coef, out if act == ACT_RELU else None, keep)
        ctx.meta = (gidx, ridx, num_bases, si, so, act, keep_scale, h_bias is not None, part, gather_input)
        ctx.pipe = (bool(gather_output), bool(x_gathered))
        ctx.w_version = weight._version
        ctx.direct = (_direct(weight), _direct(h_bias), _direct(loop_weight))
        _stamp_direct(ctx)
        return result

    @staticmethod
    def backward(ctx, grad_out):
        x_full, weight, loop_weight, coef, out, keep = ctx.saved_tensors
        gidx, ridx, nb, si, so, act, keep_scale, has_bias, part, gather_input = ctx.meta
        gather_output, x_gathered = ctx.pipe
        c, slot, row0, total = part.own_rows, part.slot_rows, part.row0, part.total_rows
        _verify_direct(ctx)
        d_w, d_b, d_l = ctx.direct
        dev, in_feat = x_full.device, x_full.shape[1]
        # (a gathered output's gradient is the consumer's reduce-scattered sums, valid in this rank's slot rows)
        grad_out = grad_out[row0:row0 + c] if gather_output else grad_out[:c]
        x_own = x_full[row0:row0 + c]
        grad_bias = grad_loop = grad_w = None
        if c == 0:           # a rank without rows: contributes zeros to the exchange
            gfull = torch.zeros(total, in_feat, dtype=torch.float32, device=dev)
            if gather_input or x_gathered:
                own = torch.empty(slot, in_feat, dtype=torch.float32, device=dev)
                if x_gathered:
                    for a, b in part.chunk_bounds():
                        part.reduce_scatter_rows(own, gfull, a, b).wait()
                    gfull[row0:row0 + slot].copy_(own)
                    return (gfull,) + (None,) * 15
                part.reduce_scatter(own, gfull).wait()
                return (own,) + (None,) * 15
            return (gfull,) + (None,) * 15
        if has_bias and ctx.needs_input_grad[2]:
            grad_bias = d_b if d_b is not None else torch.empty(grad_out.shape[1], dtype=torch.float32, device=dev)
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale, colsum_out=grad_bias, colsum_accumulate=d_b is not None)
            if d_b is not None:
                grad_bias = None
        else:
            g = epilogue_bwd(out, grad_out, act, keep, keep_scale)
        # K1^T first: its exchange then runs under the loop-weight products and the relation-weight gradient
        grad_x = pending = None
        if ctx.needs_input_grad[0]:
            pk = si * so >= 8 and pack_supported(nb, so, si, True)
            if ctx.w_bwd_packed is not None and weight._version == ctx.w_version:
                w_bwd = ctx.w_bwd_packed
            else:
                w_bwd = pack_weight(weight, nb, so, si, True) if pk else weight
            static = not gidx.sync_free and coef is not None
            coef_s, idx_s = (gidx.coef_in_src_order(coef), None) if static else (coef, gidx.by_src.perm)
            own_sum = waits = None
            if x_gathered:
                # block k = the same slot-row range of EVERY rank's slot: its partial sums are complete after its own launch and
                # leave (reduce-scatter) while block k + 1 is computed
                bounds = part.chunk_bounds()
                subs = gidx.row_blocks('src', [[(r * slot + a, r * slot + b) for r in range(part.world)] for a, b in bounds])
                gfull = torch.empty(total, in_feat, dtype=torch.float32, device=dev)
                own_sum = torch.empty(slot, in_feat, dtype=torch.float32, device=dev)
                waits = []
                for (a, b), sub in zip(bounds, subs):
                    bdd_aggregate(sub, gidx.nbr_by_src, ridx.et_by_src, coef_s, idx_s, g, w_bwd, nb, so, si, True, None,
                                  packed=pk, out=gfull)
                    waits.append(part.reduce_scatter_rows(own_sum, gfull, a, b))
            else:
                gfull = bdd_aggregate(gidx.by_src.seg, gidx.nbr_by_src, ridx.et_by_src, coef_s, idx_s, g, w_bwd, nb, so, si,
                                      True, None, packed=pk)                      # (total, in): this rank's partial sums
            if gather_input:
                grad_x = torch.empty(slot, in_feat, dtype=torch.float32, device=dev)
                pending = part.reduce_scatter(grad_x, gfull)
            else:
                grad_x = gfull
        gx_loop = None
        if loop_weight is not None:
            if ctx.needs_input_grad[3]:
                grad_loop = gemm(x_own, g, trans_a=True, split_k=pick_split_k(in_feat, g.shape[1], c), out=d_l,
                                 accumulate=d_l is not None)
                if d_l is not None:
                    grad_loop = None
            if ctx.needs_input_grad[0]:
                gx_loop = gemm(g, loop_weight, trans_b=True)
        if ctx.needs_input_grad[1]:
            static = not gidx.sync_free and coef is not None
            coef_r, idx_r = (ridx.coef_in_rel_order(coef), None) if static else (coef, ridx.by_rel.perm)
            grad_w = bdd_grad_weight(ridx.by_rel.seg, ridx.src_by_rel, ridx.dst_by_rel, coef_r, idx_r, x_full, g, nb, si,
                                     so, out=d_w, accumulate=d_w is not None)
            if d_w is not None:
                grad_w = None
        if pending is not None:
            pending.wait()
        if x_gathered and ctx.needs_input_grad[0]:
            for wt in waits:
                wt.wait()
            grad_x[row0:row0 + slot].copy_(own_sum)      # the summed gradient in this rank's slot rows; the other rows are never read
        if gx_loop is not None:       # the self-loop term only touches the rank's own rows
            own = grad_x[:c] if gather_input else grad_x[row0:row0 + c]
            lib.call('gv_axpby', own.numel(), None, 1.0, ptr(gx_loop), 1.0, ptr(own), lib.stream())
        return (grad_x, grad_w, grad_bias, grad_loop) + (None,) * 12


def rel_graph_conv_rows(x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, part, act=ACT_NONE, keep=None,
                        keep_scale=1.0, gather_input=True, pad_output=False, gather_output=False, x_gathered=False):
    return _RelGraphConvRows.apply(x, weight, h_bias, loop_weight, norm, gidx, ridx, num_bases, act, keep,
                                   float(keep_scale), part, bool(gather_input), bool(pad_output), bool(gather_output),
                                   bool(x_gathered))


class _PadRows(torch.autograd.Function):
    """(c, h) -> (rows, h) with a zero tail (a rank's slot of the row partition); backward returns the first c rows."""

    @staticmethod
    def forward(ctx, x, rows):
        ctx.c = x.shape[0]
        out = torch.empty(rows, x.shape[1], dtype=x.dtype, device=x.device)
        out[:ctx.c].copy_(x)
        if rows > ctx.c:
            out[ctx.c:].zero_()
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.c], None


def pad_rows(x, rows):
    return x if x.shape[0] == rows else _PadRows.apply(x, rows)


class _LinComb2(torch.autograd.Function):
    """wa*a + wb*b on device scalars (b may be None): the rank-local share of the loss in the row partition."""

    @staticmethod
    def forward(ctx, a, wa, b, wb):
        ctx.w = (float(wa), float(wb), b is not None)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        lib.call('gv_lincomb4', ptr(a), float(wa), ptr(b), float(wb), None, 0.0, None, 0.0, ptr(out), lib.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        wa, wb, has_b = ctx.w
        g = _chk(g.reshape(1).contiguous(), name='g')
        ga = torch.empty((), dtype=torch.float32, device=g.device)
        lib.call('gv_lincomb4', ptr(g), wa, None, 0.0, None, 0.0, None, 0.0, ptr(ga), lib.stream())
        gb = None
        if has_b:
            gb = torch.empty((), dtype=torch.float32, device=g.device)
            lib.call('gv_lincomb4', ptr(g), wb, None, 0.0, None, 0.0, None, 0.0, ptr(gb), lib.stream())
        return ga, None, gb, None


def lincomb2(a, wa, b=None, wb=0.0):
    return _LinComb2.apply(a, wa, b, wb)


class _Linear(torch.autograd.Function):
    """y = act(x @ W^T + b) with W (out, in) -- torch's F.linear layout (MaskedLinear, flow_network.py:14-15)."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, _ = _row_major(x, 'x')
        y = gemm(x, w, trans_b=True, bias=b, act=act)
        ctx.save_for_backward(x, w, y if act == ACT_RELU else None)
        ctx.act, ctx.has_bias = act, b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        g = epilogue_bwd(y, gy, ctx.act, None, 1.0) if ctx.act == ACT_RELU else _chk(gy.contiguous(), name='gy')
        gx = gemm(g, w) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = gemm(g, x, trans_a=True, split_k=pick_split_k(g.shape[1], x.shape[1], x.shape[0]))
        gb = colsum(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


def linear(x, w, b=None, act=ACT_NONE):
    return _Linear.apply(x, w, b, act)


class _Mul(torch.autograd.Function):
    """mask * weight (MaskedLinear); the mask is a constant buffer."""

    @staticmethod
    def forward(ctx, mask, w):
        ctx.save_for_backward(mask)
        ctx.direct = _direct(w)
        _stamp_direct(ctx)
        return mul(mask, w)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        _verify_direct(ctx)
        tgt = ctx.direct
        if tgt is not None and tgt.data_ptr() in GRAD_FRESH and tgt.is_contiguous():
            # the weight's slice of the optimiser arena is still all-zero: the masked gradient is written straight into it
            # (no temporary, no AccumulateGrad add)
            g = _chk(g.contiguous(), name='grad')
            lib.call('gv_mul', g.numel(), ptr(mask), ptr(g), ptr(tgt), lib.stream())
            GRAD_FRESH.discard(tgt.data_ptr())
            return None, None
        return None, mul(mask, g)


def masked_weight(mask, w):
    return _Mul.apply(mask, w)


class KLLink:
    """Hand-off between the reparameterisation node and the loss head when K3 and K6 are fused (gv_reparam_kl_fwd / _bwd):
    the forward leaves the KL pass's responsibilities and workspace here for the loss head; the loss head's backward leaves
    the upstream scalar and its two scales here and lets the reparameterisation's backward do KL's node gradients."""
    __slots__ = ('resp', 'ws', 'z_pre_ptr', 'z_pre_version', 'gkl', 'gscale', 'z_extra', 'claimed')

    def __init__(self, resp, ws, z_pre):
        self.resp, self.ws = resp, ws
        self.z_pre_ptr, self.z_pre_version = z_pre.data_ptr(), z_pre._version
        self.gkl = None
        self.gscale = self.z_extra = 0.0
        self.claimed = False          # ONE loss head may take the hand-off; a second one on the same z runs its own KL passes

    def matches(self, z_pre):
        return z_pre is not None and z_pre.data_ptr() == self.z_pre_ptr and z_pre._version == self.z_pre_version


FUSE_REPARAM_KL = _os.environ.get('GV_FUSE_REPARAM_KL', '1') == '1'


class _Reparam(torch.autograd.Function):
    """(z, m, v) = reparameterise(h2, eps): m = h2[:, :h], v = softplus(h2[:, h:]) + 1e-8, z = m + eps*sqrt(v).
    With ``z_pre`` (the mixture prior's parameters, (2k, h)) the KL forward pass over the same rows rides along."""

    @staticmethod
    def forward(ctx, h2, eps, z_pre, link_box):
        ctx.set_materialize_grads(False)
        h2 = _chk(h2.contiguous(), name='h2')
        n, h = h2.shape[0], h2.shape[1] // 2
        eps = _chk(eps.contiguous(), name='eps')
        z = torch.empty(n, h, dtype=torch.float32, device=h2.device)
        v = torch.empty_like(z)
        m = torch.empty_like(z)
        ctx.link = None
        if z_pre is not None:
            z_pre = _chk(z_pre.contiguous(), name='z_pre')
            k = z_pre.shape[0] // 2
            ws = torch.empty(int(lib.load().gv_kl_workspace_bytes(n, h, k)) // 4, dtype=torch.float32, device=h2.device)
            resp = torch.empty(n, k, dtype=torch.float32, device=h2.device)
            lib.call('gv_reparam_kl_fwd', ptr(h2), ptr(eps), ptr(z_pre), ptr(z), ptr(v), ptr(m), ptr(resp), ptr(ws), n, h, k,
                     lib.stream())
            ctx.link = link_box[0] = KLLink(resp, ws, z_pre)
            ctx.save_for_backward(h2, eps, v, z, z_pre)
        else:
            lib.call('gv_reparam_fwd', ptr(h2), ptr(eps), ptr(z), ptr(v), ptr(m), n, h, lib.stream())
            ctx.save_for_backward(h2, eps, v)
        return z, m, v

    @staticmethod
    def backward(ctx, gz, gm, gv):
        link = ctx.link
        if link is not None and link.gkl is not None:      # the loss head left KL's backward to this node
            h2, eps, v, z, z_pre = ctx.saved_tensors
            n, h = v.shape
            gz = None if gz is None else _chk(gz.contiguous(), name='gz')
            d_zp = _direct_flat(z_pre)
            gzp = d_zp if d_zp is not None else torch.empty_like(z_pre)
            gh2 = torch.empty_like(h2)
            lib.call('gv_reparam_kl_bwd', ptr(z), ptr(h2), ptr(v), ptr(eps), ptr(z_pre), ptr(link.resp), ptr(link.gkl),
                     float(link.gscale), float(link.z_extra), ptr(gz), ptr(gh2), ptr(gzp), 1 if d_zp is not None else 0,
                     ptr(link.ws), n, h, z_pre.shape[0] // 2, lib.stream())
            link.gkl = None
            if gm is not None or gv is not None:       # z_mean / z_sigma have another consumer (a second loss head, a user's own term)
                extra = torch.empty_like(h2)
                gm = None if gm is None else _chk(gm.contiguous(), name='gm')
                gv = None if gv is None else _chk(gv.contiguous(), name='gv')
                lib.call('gv_reparam_bwd', ptr(h2), ptr(eps), ptr(v), None, ptr(gm), ptr(gv), ptr(extra), n, h, lib.stream())
                gh2 += extra
            return gh2, None, (None if d_zp is not None else gzp), None
        h2, eps, v = ctx.saved_tensors[:3]
        n, h = v.shape
        gz = None if gz is None else _chk(gz.contiguous(), name='gz')
        gm = None if gm is None else _chk(gm.contiguous(), name='gm')
        gv = None if gv is None else _chk(gv.contiguous(), name='gv')
        gh2 = torch.empty_like(h2)
        lib.call('gv_reparam_bwd', ptr(h2), ptr(eps), ptr(v), ptr(gz), ptr(gm), ptr(gv), ptr(gh2), n, h, lib.stream())
        return gh2, None, None, None


def reparam(h2, eps, z_pre=None):
    """``z_pre``: when the caller knows that the loss head will evaluate KL(z) against this mixture with no flow in between,
    the KL forward pass is done in the same sweep; the returned z then carries the hand-off (``z._gv_kl_link``)."""
    if z_pre is None or not FUSE_REPARAM_KL:
        return _Reparam.apply(h2, eps, None, None)
    box = [None]
    z, m, v = _Reparam.apply(h2, eps, z_pre, box)
    z._gv_kl_link = box[0]
    return z, m, v


class _DistMultBCE(torch.autograd.Function):
    """(loss, score) of the DistMult scorer with BCE-with-logits (mean); ``bias`` = flow_log_prob or None.
    ``gscore`` handed to backward (from users of ``score``) is added to the BCE gradient."""

    @staticmethod
    def forward(ctx, embed, w_rel, bias, labels, tidx):
        ctx.set_materialize_grads(False)
        embed, ld_e = _row_major(embed, 'embed')
        w_rel, ld_w = _row_major(w_rel, 'w_relation')
        labels = _chk(labels.reshape(-1), name='labels')
        T, h = tidx.T, embed.shape[1]
        if labels.numel() != T:
            raise ValueError('labels / triplets length mismatch')
        score = torch.empty(T, dtype=torch.float32, device=embed.device)
        loss = torch.empty((), dtype=torch.float32, device=embed.device)
        ws = torch.empty(1024, dtype=torch.float32, device=embed.device)
        lib.call('gv_distmult_bce_fwd', ptr(embed), ld_e, ptr(w_rel), ld_w, ptr(tidx.trip32), ptr(tidx.fwd_order),
                 ptr(labels), ptr(bias), ptr(score), ptr(loss), ptr(ws), T, h, lib.stream())
        ctx.save_for_backward(embed, w_rel, labels, score)
        ctx.tidx, ctx.has_bias = tidx, bias is not None
        ctx.mark_non_differentiable(score)
        return loss, score

    @staticmethod
    def backward(ctx, gloss, _gscore):
        if gloss is None:
            return None, None, None, None, None
        embed, w_rel, labels, score = ctx.saved_tensors
        tidx = ctx.tidx
        T, h = tidx.T, embed.shape[1]
        dev = embed.device
        gloss = _chk(gloss.reshape(1).contiguous(), name='gloss')
        dscore = torch.empty(T, dtype=torch.float32, device=dev)
        dbias = torch.empty((), dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = torch.empty(1024, dtype=torch.float32, device=dev)
        lib.call('gv_bce_grad', ptr(score), ptr(labels), ptr(gloss), ptr(dscore), None, None, None, ptr(dbias), ptr(ws), T,
                 lib.stream())
        g_embed = g_w = None
        if ctx.needs_input_grad[0]:
            g_embed = bdd_aggregate(tidx.inc, tidx.inc_other, tidx.inc_rel, dscore, tidx.inc_tid, embed, w_rel,
                                    h, 1, 1)
        if ctx.needs_input_grad[1]:
            g_w = bdd_grad_weight(tidx.rel, tidx.rel_s, tidx.rel_o, dscore, tidx.rel_tid, embed, embed, h, 1, 1)
        return g_embed, g_w, dbias, None, None


def distmult_bce(embed, w_rel, bias, labels, tidx):
    return _DistMultBCE.apply(embed, w_rel, bias, labels, tidx)


class _DistMultScore(torch.autograd.Function):
    """score_t = sum_d e[s,d] w[r,d] e[o,d]  (LinkPredict.calc_score) with a general backward."""

    @staticmethod
    def forward(ctx, embed, w_rel, tidx):
        embed, ld_e = _row_major(embed, 'embed')
        w_rel, ld_w = _row_major(w_rel, 'w_relation')
        T, h = tidx.T, embed.shape[1]
        score = torch.empty(T, dtype=torch.float32, device=embed.device)
        loss = torch.empty((), dtype=torch.float32, device=embed.device)
        ws = torch.empty(1024, dtype=torch.float32, device=embed.device)
        zeros = torch.zeros(T, dtype=torch.float32, device=embed.device)
        lib.call('gv_distmult_bce_fwd', ptr(embed), ld_e, ptr(w_rel), ld_w, ptr(tidx.trip32), ptr(tidx.fwd_order), ptr(zeros),
                 None, ptr(score), ptr(loss), ptr(ws), T, h, lib.stream())
        ctx.save_for_backward(embed, w_rel)
        ctx.tidx = tidx
        return score

    @staticmethod
    def backward(ctx, gscore):
        embed, w_rel = ctx.saved_tensors
        tidx, h = ctx.tidx, embed.shape[1]
        gscore = _chk(gscore.contiguous(), name='gscore')
        g_embed = g_w = None
        if ctx.needs_input_grad[0]:
            g_embed = bdd_aggregate(tidx.inc, tidx.inc_other, tidx.inc_rel, gscore, tidx.inc_tid, embed, w_rel,
                                    h, 1, 1)
        if ctx.needs_input_grad[1]:
            g_w = bdd_grad_weight(tidx.rel, tidx.rel_s, tidx.rel_o, gscore, tidx.rel_tid, embed, embed, h, 1, 1)
        return g_embed, g_w, None


def distmult_score(embed, w_rel, tidx):
    return _DistMultScore.apply(embed, w_rel, tidx)


class _MeanSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x.contiguous(), name='x')
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = torch.empty(1024, dtype=torch.float32, device=x.device)
        lib.call('gv_mean_sq', ptr(x), x.numel(), 1.0 / x.numel(), ptr(out), ptr(ws), 0, lib.stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return axpby(2.0 / x.numel(), x, a=_chk(g.reshape(1).contiguous(), name='g'))


def mean_sq(x):
    return _MeanSq.apply(x)


class _KL(torch.autograd.Function):
    """KGVAE.get_kl: mean_n[ logN(z; m, v) + flp - log mean_j N(z; m_j, v_j) ]; z_pre is (2k, h)."""

    @staticmethod
    def forward(ctx, z, m, v, z_pre, flp):
        z, m, v = (_chk(t.contiguous(), name=nm) for t, nm in ((z, 'z'), (m, 'z_mean'), (v, 'z_sigma')))
        z_pre = _chk(z_pre.contiguous(), name='z_pre')
        n, h = z.shape
        k = z_pre.shape[0] // 2
        ws = torch.empty(int(lib.load().gv_kl_workspace_bytes(n, h, k)) // 4, dtype=torch.float32, device=z.device)
        resp = torch.empty(n, k, dtype=torch.float32, device=z.device)
        kl = torch.empty((), dtype=torch.float32, device=z.device)
        lib.call('gv_kl_fwd', ptr(z), ptr(m), h, ptr(v), ptr(z_pre), ptr(flp), ptr(resp), ptr(kl), ptr(ws), n, h, k,
                 None, lib.stream())
        ctx.save_for_backward(z, m, v, z_pre, resp, ws)
        ctx.has_flp = flp is not None
        ctx.zp_version = z_pre._version
        return kl

    @staticmethod
    def backward(ctx, gkl):
        z, m, v, z_pre, resp, ws = ctx.saved_tensors       # ws still holds the forward's mixture table
        n, h = z.shape
        k = z_pre.shape[0] // 2
        gkl = _chk(gkl.reshape(1).contiguous(), name='gkl')
        gz, gm, gv = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        d_zp = _direct_flat(z_pre)
        gzp = d_zp if d_zp is not None else torch.empty_like(z_pre)
        lib.call('gv_kl_bwd', ptr(z), ptr(m), h, ptr(v), ptr(z_pre), ptr(resp), ptr(gkl), 1.0, 0.0, ptr(gz), ptr(gm), ptr(gv),
                 ptr(gzp), 1 if d_zp is not None else 0, ptr(ws), 1, n, h, k, None, lib.stream())
        return gz, gm, gv, (None if d_zp is not None else gzp), (copy_of(gkl.reshape(1)).reshape(()) if ctx.has_flp else None)


def kl_to_mixture(z, m, v, z_pre, flp=None):
    return _KL.apply(z, m, v, z_pre, flp)


class _IAFUpdate(torch.autograd.Function):
    """x_new[:, c] = z*exp(alpha+mu) where colcount[c] > 0 else x_old; net = [mu | alpha]."""

    @staticmethod
    def forward(ctx, z, net, x_old, colcount):
        z, net, x_old = (_chk(t.contiguous(), name=nm) for t, nm in ((z, 'z'), (net, 'net'), (x_old, 'x_old')))
        n, d = z.shape
        x_new = torch.empty_like(z)
        lib.call('gv_iaf_update_fwd', ptr(z), ptr(net), 2 * d, ptr(x_old), ptr(colcount), ptr(x_new), n, d, lib.stream())
        ctx.save_for_backward(z, net, colcount)
        return x_new

    @staticmethod
    def backward(ctx, gx):
        z, net, colcount = ctx.saved_tensors
        n, d = z.shape
        gx = _chk(gx.contiguous(), name='gx')
        gz, gnet, gold = torch.empty_like(z), torch.empty_like(net), torch.empty_like(z)
        lib.call('gv_iaf_update_bwd', ptr(z), ptr(net), 2 * d, ptr(colcount), ptr(gx), None, ptr(gz), ptr(gnet), ptr(gold),
                 n, d, lib.stream())
        return gz, gnet, gold, None


MADE_ROW0_BWD = _os.environ.get('GV_MADE_ROW0_BWD', '1') == '1'


def iaf_bwd_row0(z, net_row, colcount0, g_cur, gld, g_z):
    """Pass 0 of a MADE backward (the update was fed one broadcast [mu | alpha] row): g_z += the pass's share, returns the
    gradient w.r.t. that row (1, 2d) -- gv_iaf_update_bwd_row0, or the generic update backward + column sums + axpby."""
    n, d = z.shape
    st = lib.stream()
    if MADE_ROW0_BWD and d % 4 == 0 and d <= 1024 and net_row.is_contiguous():
        ws = torch.empty(int(lib.load().gv_iaf_update_bwd_row0_workspace_floats(d)), dtype=torch.float32, device=z.device)
        g_row = torch.empty(1, 2 * d, dtype=torch.float32, device=z.device)
        lib.call('gv_iaf_update_bwd_row0', ptr(z), ptr(net_row), ptr(colcount0), ptr(g_cur), ptr(gld), ptr(g_z), ptr(g_row), ptr(ws),
                 n, d, st)
        return g_row
    gz_p = torch.empty(n, d, dtype=torch.float32, device=z.device)
    g_net0 = torch.empty(n, 2 * d, dtype=torch.float32, device=z.device)
    g_dump = torch.empty(n, d, dtype=torch.float32, device=z.device)
    lib.call('gv_iaf_update_bwd', ptr(z), ptr(net_row), 0, ptr(colcount0), ptr(g_cur), ptr(gld), ptr(gz_p), ptr(g_net0), ptr(g_dump),
             n, d, st)
    lib.call('gv_axpby', n * d, None, 1.0, ptr(gz_p), 1.0, ptr(g_z), st)
    return colsum(g_net0).view(1, -1)


def iaf_update(z, net, x_old, colcount):
    return _IAFUpdate.apply(z, net, x_old, colcount)


class _RowSumCols(torch.autograd.Function):
    """sum over columns [col0, col0+ncols) of each row (log_det = sum_d alpha)."""

    @staticmethod
    def forward(ctx, x, col0, ncols):
        x = _chk(x.contiguous(), name='x')
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        lib.call('gv_rowsum', ptr(x), x.shape[1], col0, ncols, ptr(out), x.shape[0], lib.stream())
        ctx.meta = (tuple(x.shape), col0, ncols)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, col0, ncols = ctx.meta
        gx = torch.zeros(shape, dtype=torch.float32, device=g.device)
        gx[:, col0:col0 + ncols] = g.unsqueeze(1)
        return gx, None, None


def rowsum_cols(x, col0, ncols):
    return _RowSumCols.apply(x, col0, ncols)


class _MeanRowsMulti(torch.autograd.Function):
    """mean over the rows that exist of the sum of up to 8 per-row vectors: KGVAE.flow_log_prob (kgvae/model.py:116-123) in one
    launch instead of adds + mask + sum + divide through torch; backward: one launch, the same gradient vector for every input.
    Only the first ``n`` entries of the vectors take part (rows riding along behind them -- get_mmd's prior rows -- get zero)."""

    @staticmethod
    def forward(ctx, rows_dev, n, *xs):
        xs = [_chk(x.contiguous().reshape(-1), name='log_det') for x in xs]
        length = xs[0].numel()
        if any(x.numel() != length for x in xs) or not 0 < n <= length:
            raise ValueError('mean_rows_multi: vectors of one length >= n')
        out = torch.empty((), dtype=torch.float32, device=xs[0].device)
        tab = (_ct.c_void_p * len(xs))(*[ptr(x) for x in xs])
        ws = torch.empty(64, dtype=torch.float32, device=xs[0].device)
        lib.call('gv_mean_rows_multi', len(xs), _ct.addressof(tab), n, ptr(rows_dev), ptr(out), ptr(ws), lib.stream())
        ctx.rows_dev, ctx.n, ctx.len, ctx.k = rows_dev, n, length, len(xs)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _chk(g.contiguous().reshape(()), name='g')
        gx = torch.empty(ctx.len, dtype=torch.float32, device=g.device)
        lib.call('gv_mean_rows_bwd', ptr(g), ctx.len, ctx.n, ptr(ctx.rows_dev), ptr(gx), lib.stream())
        return (None, None) + (gx,) * ctx.k


def mean_rows_multi(xs, n=None, rows_dev=None):
    """mean_r sum_i xs[i][r] over the first n rows (None: all) that exist (rows_dev: device int32 count, None = all n)."""
    xs = list(xs)
    if not 1 <= len(xs) <= 8:
        raise ValueError('mean_rows_multi: 1..8 vectors')
    return _MeanRowsMulti.apply(rows_dev, int(xs[0].numel() if n is None else n), *xs)


class _ReverseCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x.contiguous(), name='x')
        out = torch.empty_like(x)
        lib.call('gv_reverse_cols', ptr(x), ptr(out), x.shape[0], x.shape[1], lib.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        g = _chk(g.contiguous(), name='g')
        out = torch.empty_like(g)
        lib.call('gv_reverse_cols', ptr(g), ptr(out), g.shape[0], g.shape[1], lib.stream())
        return out


def reverse_cols(x):
    return _ReverseCols.apply(x)


class _MatMul(torch.autograd.Function):
    """Plain differentiable a @ b on the f32 MFMA GEMM (basis combination W = w_comp @ V)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return gemm(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _chk(g.contiguous(), name='g')
        ga = gemm(g, b, trans_b=True) if ctx.needs_input_grad[0] else None
        gb = gemm(a, g, trans_a=True, split_k=pick_split_k(a.shape[1], g.shape[1], a.shape[0])) \
            if ctx.needs_input_grad[1] else None
        return ga, gb


def matmul(a, b):
    return _MatMul.apply(a, b)


class _MMD(torch.autograd.Function):
    """KGVAE.get_mmd's three RBF-kernel means on x (sx, h) and y (sy, h), fused forward / backward."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _chk(x.contiguous(), name='z_pri'), _chk(y.contiguous(), name='z_post')
        out = torch.empty((), dtype=torch.float32, device=x.device)
        ws = torch.empty(x.shape[0] + y.shape[0], dtype=torch.float32, device=x.device)
        lib.call('gv_mmd_fwd', ptr(x), ptr(y), None, x.shape[0], y.shape[0], x.shape[1], ptr(out), ptr(ws), lib.stream())
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _chk(g.reshape(1).contiguous(), name='g')
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        lib.call('gv_mmd_bwd', ptr(x), ptr(y), None, x.shape[0], y.shape[0], x.shape[1], ptr(g), 1.0, ptr(gx), ptr(gy),
                 lib.stream())
        return gx, gy


def mmd(x, y):
    return _MMD.apply(x, y)


class _PriorSample(torch.autograd.Function):
    """z_pri = cat([m]*repeat) + eps * cat([sqrt(softplus(raw)+1e-8)]*repeat) for z_pre = [m; raw] (2k, h)."""

    @staticmethod
    def forward(ctx, z_pre, eps):
        z_pre, eps = _chk(z_pre.contiguous(), name='z_pre'), _chk(eps.contiguous(), name='eps')
        k, h = z_pre.shape[0] // 2, z_pre.shape[1]
        out = torch.empty_like(eps)
        ctx.set_materialize_grads(False)
        lib.call('gv_prior_sample_fwd', ptr(z_pre), ptr(eps), ptr(out), eps.shape[0], k, h, lib.stream())
        ctx.save_for_backward(z_pre, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        z_pre, eps = ctx.saved_tensors
        if g is None:
            return None, None
        g = _chk(g.contiguous(), name='g')
        d_zp = _direct_flat(z_pre)
        gz = d_zp if d_zp is not None else torch.empty_like(z_pre)
        lib.call('gv_prior_sample_bwd', ptr(z_pre), ptr(eps), ptr(g), ptr(gz), 1 if d_zp is not None else 0, eps.shape[0],
                 z_pre.shape[0] // 2, z_pre.shape[1], lib.stream())
        return (None if d_zp is not None else gz), None


def prior_sample(z_pre, eps):
    return _PriorSample.apply(z_pre, eps)


class _LossHead(torch.autograd.Function):
    """LinkPredict.get_loss as ONE autograd node (kgvae/link_predict.py:71-92):

        loss = BCE(DistMult(z, w_rel, triplets) + flp) + reg*(mean z^2 + mean w_rel^2) + kl_w*KL + mmd_w*MMD

    with KL = KGVAE.get_kl(z) and MMD = KGVAE.get_mmd's kernel means on (z_pri, z[pick]).  Forward is the
    K5/K6 kernels plus one scalar combine; backward produces a single gradient for z: KL's gz, the regulariser
    and the MMD rows are accumulated into one buffer that the DistMult gather-aggregate (K1, 1x1 blocks)
    takes as its fused addend -- no multi-use gradient adds, no scalar glue kernels.  The KL and MMD kernel
    groups (small, latency-bound) run on side streams beside the bandwidth-bound DistMult kernels.
    Returns (loss, predict_loss, kl, mmd); only ``loss`` is differentiable.
    """

    @staticmethod
    def forward(ctx, z, z_mean, z_sigma, w_rel, z_pre, flp, z_pri, pick, labels, tidx, reg_w, kl_w, mmd_w, score_bias,
                embed_rows=None, rows_dev=None, kl_link=None):
        z, ld_z = _row_major(z, 'embed')
        if rows_dev is not None:          # static-shape batch: rows [*rows_dev, n) of z are padding (include/gcnvae.h, rows_dev)
            rows_dev = _chk(rows_dev.reshape(-1), torch.int32, 'rows_dev')
            if kl_w <= 0 or embed_rows is not None:
                raise NotImplementedError('rows_dev needs kl_w > 0 (the padding rows are masked in the KL pass) and no embed_rows')
        w_rel, ld_w = _row_major(w_rel, 'w_relation')
        labels = _chk(labels.reshape(-1), name='labels')
        dev, (n, h), T = z.device, z.shape, tidx.T
        if labels.numel() != T:
            raise ValueError('labels / triplets length mismatch')
        f32 = dict(dtype=torch.float32, device=dev)
        ctx.set_materialize_grads(False)
        scal = torch.empty(4, **f32)                 # pred, reg, kl, mmd (each written before it is read)
        pred, reg, kl, mmd = scal[0:1], scal[1:2], scal[2:3], scal[3:4]
        ws, ws2 = torch.empty(1024, **f32), torch.empty(1024, **f32)
        score = torch.empty(T, **f32)
        loss = torch.empty((), **f32)
        bias = flp if (score_bias and flp is not None) else None
        resp = wsk = z_post = wsm = None
        link = kl_link if (kl_link is not None and not kl_link.claimed and kl_w > 0 and flp is None and rows_dev is None
                           and embed_rows is None and kl_link.matches(z_pre) and kl_link.resp.shape[0] == n) else None
        if link is not None:
            link.claimed = True
        if kl_w > 0:
            z_mean, z_sigma = _chk(z_mean.contiguous(), name='z_mean'), _chk(z_sigma.contiguous(), name='z_sigma')
            z_pre = _chk(z_pre.contiguous(), name='z_pre')
            k = z_pre.shape[0] // 2
            if link is None:
                wsk = torch.empty(int(lib.load().gv_kl_workspace_bytes(n, h, k)) // 4, **f32)
                resp = torch.empty(n, k, **f32)
        if mmd_w > 0:
            z_pri = _chk(z_pri.contiguous(), name='z_pri')
            pick = _chk(pick.reshape(-1), torch.int64, 'pick')
            wsm = torch.empty(z_pri.shape[0] + pick.numel(), **f32)
        if ld_z != h or ld_w != h:
            raise ValueError('loss_head needs contiguous embeddings and relation table')
        st = lib.stream()
        # every term leaves its per-block partial sums in its workspace; ONE combine launch finishes the four sums
        if link is not None:
            resp, wsk = link.resp, link.ws                 # the KL pass already ran with the reparameterisation
        elif kl_w > 0:
            lib.call('gv_kl_fwd', ptr(z), ptr(z_mean), h, ptr(z_sigma), ptr(z_pre), ptr(flp), ptr(resp), None, ptr(wsk),
                     n, h, k, ptr(rows_dev), st)
        if mmd_w > 0:      # the posterior sample set is rows `pick` of z, read in place
            lib.call('gv_mmd_fwd', ptr(z_pri), ptr(z), ptr(pick), z_pri.shape[0], pick.numel(), h, None, ptr(wsm), st)
        # embed_rows: the regulariser's mean runs over that many rows (the rest of z are all-zero padding rows of the
        # multi-GPU row partition)
        z_count = z.numel() if embed_rows is None else int(embed_rows) * h
        lib.call('gv_mean_sq2', ptr(z), z.numel(), 1.0 / z_count, ptr(w_rel), w_rel.numel(), 1.0 / w_rel.numel(), None,
                 ptr(ws2), ptr(rows_dev), n, st)
        # DistMult scorer + BCE (three 800-B row gathers per triplet: the bandwidth-bound part)
        lib.call('gv_distmult_bce_fwd', ptr(z), ld_z, ptr(w_rel), ld_w, ptr(tidx.trip32), ptr(tidx.fwd_order), ptr(labels),
                 ptr(bias), ptr(score), None, ptr(ws), T, h, st)
        lib.call('gv_loss_combine', ptr(ws), T, ptr(ws2), z.numel(), w_rel.numel(), ptr(wsk) if kl_w > 0 else None, n, h,
                 (z_pre.shape[0] // 2) if kl_w > 0 else 0, ptr(wsm) if mmd_w > 0 else None,
                 z_pri.shape[0] if mmd_w > 0 else 0, pick.numel() if mmd_w > 0 else 0, float(reg_w), float(kl_w),
                 float(mmd_w), ptr(scal), ptr(loss), ptr(rows_dev), st)
        ctx.save_for_backward(z, z_mean if kl_w > 0 else None, z_sigma if kl_w > 0 else None, w_rel,
                              z_pre if kl_w > 0 else None, resp, z_pri if mmd_w > 0 else None, z_post,
                              pick if mmd_w > 0 else None, labels, score, wsk, rows_dev)
        ctx.meta = (tidx, float(reg_w), float(kl_w), float(mmd_w), bias is not None, flp is not None and kl_w > 0)
        ctx.z_count = z_count
        ctx.kl_link = link
        ctx.direct_w = _direct(w_rel) if ld_w == h else None
        _stamp_direct(ctx)
        out_pred, out_kl, out_mmd = pred.reshape(()), kl.reshape(1), mmd.reshape(())
        ctx.mark_non_differentiable(out_pred, out_kl, out_mmd)
        return loss, out_pred, out_kl, out_mmd

    @staticmethod
    def backward(ctx, g, _gp, _gk, _gm):
        if g is None:
            return (None,) * 17
        z, z_mean, z_sigma, w_rel, z_pre, resp, z_pri, z_post, pick, labels, score, wsk, rows_dev = ctx.saved_tensors
        z_count = ctx.z_count
        tidx, reg_w, kl_w, mmd_w, has_bias, flp_in_kl = ctx.meta
        dev, (n, h), T = z.device, z.shape, tidx.T
        f32 = dict(dtype=torch.float32, device=dev)
        g = _chk(g.reshape(1).contiguous(), name='gloss')
        # every buffer a side branch writes is allocated here, before the forks
        dscore = torch.empty(T, **f32)
        dbias = torch.empty((), **f32) if has_bias else None      # (written by gv_bce_grad's ordered final sum: no fill)
        ws = torch.empty(1024, **f32)
        link = ctx.kl_link
        gz = torch.empty_like(z) if link is None else None
        gm = gv = gzp = d_zp = g_pri = g_post = None
        if link is not None:
            # KL's node gradients and the regulariser's are chained through the reparameterisation by ITS backward
            # (gv_reparam_kl_bwd): here only the scalar and the two scales are handed over
            link.gkl, link.gscale, link.z_extra = g, kl_w, 2.0 * reg_w / z_count
            d_zp = True          # z_pre's gradient is produced there too
        elif kl_w > 0:
            k = z_pre.shape[0] // 2                       # wsk: the forward's workspace, mixture table still valid
            d_zp = _direct_flat(z_pre)
            gm, gv = torch.empty_like(z), torch.empty_like(z)
            gzp = d_zp if d_zp is not None else torch.empty_like(z_pre)
        if mmd_w > 0:
            g_pri = torch.empty_like(z_pri)
        _verify_direct(ctx)
        d_w = ctx.direct_w
        g_w = d_w if d_w is not None else torch.empty_like(w_rel)
        g_flp = torch.empty((), **f32) if (has_bias or flp_in_kl) else None
        st = lib.stream()
        # branch 1: KL backward (writes gz, gm, gv, gzp), then the regulariser folded into gz
        with fork(1):
            s1 = lib.stream()
            if link is not None:
                pass
            elif kl_w > 0:
                # the embedding regulariser's gradient g * (2 reg_w / numel) * z rides on the same pass over z
                lib.call('gv_kl_bwd', ptr(z), ptr(z_mean), h, ptr(z_sigma), ptr(z_pre), ptr(resp), ptr(g), kl_w,
                         2.0 * reg_w / z_count, ptr(gz), ptr(gm), ptr(gv), ptr(gzp), 1 if d_zp is not None else 0, ptr(wsk),
                         1, n, h, k, ptr(rows_dev), s1)
            else:
                lib.call('gv_axpby', z.numel(), ptr(g), 2.0 * reg_w / z_count, ptr(z), 0.0, ptr(gz), s1)
            if mmd_w > 0 and link is None:      # MMD backward: prior rows -> g_pri, posterior rows ADDED into rows `pick` of gz (now complete)
                lib.call('gv_mmd_bwd', ptr(z_pri), ptr(z), ptr(pick), z_pri.shape[0], pick.numel(), h, ptr(g), mmd_w,
                         ptr(g_pri), ptr(gz), s1)
        # main: dL/dscore, then the relation-side gradient (does not need gz)
        if tidx.pos3 is not None:      # dL/dscore also lands in the two launches' own orders
            d_inc, d_rel, idx_inc, idx_rel = torch.empty(2 * T, **f32), torch.empty(T, **f32), None, None
        else:
            d_inc, d_rel, idx_inc, idx_rel = dscore, dscore, tidx.inc_tid, tidx.rel_tid
        lib.call('gv_bce_grad', ptr(score), ptr(labels), ptr(g), ptr(dscore), ptr(tidx.pos3),
                 ptr(d_inc) if tidx.pos3 is not None else None, ptr(d_rel) if tidx.pos3 is not None else None,
                 ptr(dbias), ptr(ws), T, st)
        bdd_grad_weight(tidx.rel, tidx.rel_s, tidx.rel_o, d_rel, idx_rel, z, z, h, 1, 1, out=g_w,
                        accumulate=d_w is not None)
        lib.call('gv_axpby', w_rel.numel(), ptr(g), 2.0 * reg_w / w_rel.numel(), ptr(w_rel), 1.0, ptr(g_w), st)
        join(1)
        g_z = bdd_aggregate(indices.largest_first(tidx.inc) if indices.K1_ITEMS_LARGEST_FIRST else tidx.inc, tidx.inc_other, tidx.inc_rel,
                            d_inc, idx_inc, z, w_rel, h, 1, 1, addend=gz)
        if link is not None and mmd_w > 0:       # no KL buffer to ride on: the MMD rows are added to the decoder's gradient
            lib.call('gv_mmd_bwd', ptr(z_pri), ptr(z), ptr(pick), z_pri.shape[0], pick.numel(), h, ptr(g), mmd_w,
                     ptr(g_pri), ptr(g_z), st)
        if g_flp is not None:
            lib.call('gv_lincomb4', ptr(dbias) if has_bias else None, 1.0, ptr(g) if flp_in_kl else None, kl_w, None, 0.0,
                     None, 0.0, ptr(g_flp), st)
        return (g_z, gm, gv, (None if d_w is not None else g_w), (None if d_zp is not None else gzp), g_flp, g_pri, None,
                None, None, None, None, None, None, None, None, None)


def loss_head(z, z_mean, z_sigma, w_rel, z_pre, flp, z_pri, pick, labels, tidx, reg_w, kl_w, mmd_w, score_bias,
              embed_rows=None, rows_dev=None):
    return _LossHead.apply(z, z_mean, z_sigma, w_rel, z_pre, flp, z_pri, pick, labels, tidx, float(reg_w), float(kl_w),
                           float(mmd_w), bool(score_bias), embed_rows, rows_dev, getattr(z, '_gv_kl_link', None))


# ------------------------------------------------------------------------------------------------
# K4, the masked MLP of the IAF blocks (MADE): made.py -- same namespace for the callers (ops.made_forward, ops.made_chain, ...)
# ------------------------------------------------------------------------------------------------
from .made import *                                              # noqa: E402,F401,F403
from .made import _ChainLayer, _RowLayer, _MADEForward, _MADEForwardBF16      # noqa: E402,F401
from . import made                                               # noqa: E402,F401  (ops.made.<KNOB>)
