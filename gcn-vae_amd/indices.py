"""The index side of K1: work-item lists over sorted edge lists (SegmentItems), the per-graph orderings and their caches
(GraphIndex: by destination / by source; RelationIndex: by relation, relation-sorted rows, relation phases; TripletIndex: the decoder's
incidence lists), built on the device by the entry points of csrc/k_index.hip, and the two K1 launch forms that need an index of
their own: the relation-phase kernel (PhaseOrder, bdd_aggregate_phases: csrc/k_phase.hip) and the LDS-resident kernel (LdsOrder,
bdd_aggregate_lds: csrc/k_lds.hip).

Part of the ``ops`` namespace (ops re-exports everything here); the knobs of this file (K1_PHASES, PHASE_*, K1_LDS*, NATIVE_INDEX)
are THIS module's globals.
"""
import ctypes as _ct
import itertools as _it
import os as _os
from dataclasses import dataclass
from typing import Optional

import torch

from . import lib
from . import ops as _ops
from .lib import ptr
from .ops import ACT_NONE, DEFAULT_CHUNK, DEFAULT_CHUNK_REL, _chk, _row_major, chunk_for


# ------------------------------------------------------------------------------------------------
# segment work items
@dataclass
class SegmentItems:
    items: torch.Tensor      # int32 [n_items, 4]
    fix: torch.Tensor        # int32 [n_fix, 4]
    n_items: int
    n_fix: int
    n_slots: int
    rowptr: torch.Tensor     # int32 [n_seg + 1]
    chunk: int
    exact: bool = True       # False: sized by upper bounds and -1 padded (the synchronisation-free builders of per-batch indices)


def build_segment_items(rowptr: torch.Tensor, chunk: int, n_edges: Optional[int] = None) -> SegmentItems:
    """Cut a CSR segment list into <=chunk-edge work items (two device kernels around one scan).

    ``n_edges`` None: exact lists; the three totals come back to the host (ONE synchronisation) -- right for an index
    that is built once per graph.  ``n_edges`` given (= rowptr[-1], known to the caller): no synchronisation at all --
    the lists are sized by their upper bounds (items <= n_seg + E/chunk, fix-ups <= min(n_seg, E/chunk), slots <=
    2 E/chunk) and pre-filled with -1, which the kernels skip -- right for per-mini-batch indices."""
    _chk(rowptr, torch.int32, 'rowptr')
    n_seg = rowptr.numel() - 1
    dev = rowptr.device
    counts = torch.empty(3, n_seg, dtype=torch.int32, device=dev)        # n_chunks, n_slots, is_split
    lib.call('gv_segment_items_count', ptr(rowptr), n_seg, chunk, ptr(counts[0]), ptr(counts[1]), ptr(counts[2]),
             lib.stream())
    offs = torch.zeros(3, n_seg + 1, dtype=torch.int32, device=dev)      # exclusive scans
    offs[:, 1:] = torch.cumsum(counts, 1, dtype=torch.int32)
    if n_edges is None:
        n_items, n_slot_total, n_fix = (int(v) for v in offs[:, -1].tolist())
        items = torch.empty(max(n_items, 1), 4, dtype=torch.int32, device=dev)
        fix = torch.empty(max(n_fix, 1), 4, dtype=torch.int32, device=dev)
    else:
        extra = int(n_edges) // chunk + 1
        n_items, n_fix, n_slot_total = n_seg + extra, min(n_seg, extra), 2 * extra
        items = torch.full((max(n_items, 1), 4), -1, dtype=torch.int32, device=dev)
        fix = torch.full((max(n_fix, 1), 4), -1, dtype=torch.int32, device=dev)
    lib.call('gv_segment_items_fill', ptr(rowptr), n_seg, chunk, ptr(offs[0]), ptr(offs[1]), ptr(offs[2]),
             ptr(items), ptr(fix), lib.stream())
    return SegmentItems(items, fix, n_items, n_fix, n_slot_total, rowptr, chunk, exact=n_edges is None)


# Work items of a static graph's K1 launches in order of their length (longest first).  A workgroup is four waves = four consecutive
# items and ends with its longest: in the built (row) order a 128-edge piece of a hub row sits beside rows of 3 edges.  Sorted, a
# workgroup's four items are equally long and the launch ends on short ones.  Measured on the default configuration (serialised kernel
# trace): aggregations 107.2 -> 102.8, 82.9 -> 80.3, 60.0 -> 57.3, 61.4 -> 58.5, 41.2 -> 38.3 us, step 1.025 -> 1.00-1.01 ms; moving only the
# long items to the front gains nothing (it is the waves of a workgroup, not the launch's tail); the grad-W lists (XCD windows) the same
# either way.  Results are bit-identical (an item's contents do not change).  GV_K1_ITEMS_LPT=0: the built order.
K1_ITEMS_LARGEST_FIRST = _os.environ.get('GV_K1_ITEMS_LPT', '1') == '1'


def largest_first(seg: SegmentItems) -> SegmentItems:
    """The same work items, the long ones (the <= chunk-edge pieces of hub rows) FIRST, ties in the built order: the launch's last
    workgroups are then short ones (longest-processing-time-first list scheduling; rows keep their items' contents, so results are
    bit-identical).  Cached on the list; lists of a per-batch index (-1 padded, rebuilt every step) are returned as they are."""
    hit = getattr(seg, '_largest_first', None)
    if hit is not None:
        return hit
    n = seg.n_items
    if n < 1024 or not seg.exact:
        seg._largest_first = seg
        return seg
    it = seg.items[:n]
    order = torch.argsort((it[:, 2] - it[:, 1]).long(), descending=True, stable=True)
    items = seg.items.clone()
    items[:n] = it[order]
    out = SegmentItems(items, seg.fix, seg.n_items, seg.n_fix, seg.n_slots, seg.rowptr, seg.chunk)
    out._largest_first = out
    seg._largest_first = out
    return out


def _rowptr_from_sorted(keys_sorted: torch.Tensor, n_seg: int) -> torch.Tensor:
    """CSR pointer of a sorted key list (no host synchronisation: torch.bincount would size its output on the host)."""
    bounds = torch.arange(n_seg + 1, device=keys_sorted.device, dtype=keys_sorted.dtype)
    return torch.searchsorted(keys_sorted.contiguous(), bounds).to(torch.int32)


def _index_caps(n_entries: int, n_seg: int, chunk: int):
    """Upper bounds of the work-item lists of one ordering (= gv_index_caps): items, fix-ups, partial-row slots."""
    extra = int(n_entries) // chunk + 1
    return n_seg + extra, max(1, min(n_seg, extra)), 2 * extra


def _carve_i32(device, sizes):
    """One int32 allocation cut into 256-B aligned pieces (a native index build writes ~10 arrays: one malloc)."""
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (int(n) + 63) // 64 * 64
    arena = torch.empty(max(total, 64), dtype=torch.int32, device=device)
    return [arena[o:o + int(n)] for o, n in zip(offs, sizes)]


def _index_workspace(device, n_entries, n_seg_max):
    nbytes = int(lib.load().gv_index_workspace_bytes(int(n_entries), int(n_seg_max)))
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


NATIVE_INDEX = _os.environ.get('GV_NATIVE_INDEX', '1') == '1'     # sync-free indices through gv_*_index_build (one C call each)


@dataclass
class EdgeOrder:
    """One ordering of the edge list: the segment key (dst, src or relation) is sorted."""
    perm: Optional[torch.Tensor]   # int32 [E] original edge id of each position; None = identity
    seg: SegmentItems


class GraphIndex:
    """Device-side index of a relational graph for the K1 kernels.

    by_dst : CSR over destinations  (forward aggregation; the reference's edge order is already
             dst-sorted, kgvae/utils.py:146-147, so ``perm`` is usually None)
    by_src : CSC over sources       (backward w.r.t. x)
    Relation-dependent arrays live in ``RelationIndex`` (etypes arrive per forward call).
    """

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, chunk: Optional[int] = None,
                 dst_sorted: Optional[bool] = None, sync_free: bool = False, num_src_nodes: Optional[int] = None):
        """``dst_sorted``: None = check (one host synchronisation), True = the caller guarantees dst is non-decreasing.
        ``sync_free``: size the work-item lists by upper bounds instead of reading their totals back (per-batch graphs).
        ``num_src_nodes``: a RECTANGULAR graph -- destinations index ``num_nodes`` rows (a rank's own row block of the
        multi-GPU destination-row partition), sources index a table of ``num_src_nodes`` rows (all nodes)."""
        if not src.is_cuda:
            raise RuntimeError('GraphIndex needs CUDA index tensors; there is no CPU fallback')
        self.num_nodes, self.num_edges = int(num_nodes), int(src.numel())
        chunk = chunk_for(self.num_edges) if chunk is None else int(chunk)
        self.num_src_nodes = self.num_nodes if num_src_nodes is None else int(num_src_nodes)
        self.device = src.device
        self.sync_free = bool(sync_free)
        ne = self.num_edges if sync_free else None
        self._rel_cache = {}
        self._chunk_cache = {}
        self._lds_seg_cache = {}
        if sync_free and NATIVE_INDEX:
            self._build_native(src, dst, chunk, bool(dst_sorted))
            return
        src = src.to(torch.int64)
        dst = dst.to(torch.int64)
        self.src32, self.dst32 = src.to(torch.int32), dst.to(torch.int32)
        if dst_sorted is None:
            dst_sorted = bool(self.num_edges) and bool((dst[1:] >= dst[:-1]).all())
        if dst_sorted:
            perm_d = None
            dst_keys = dst
        else:
            perm_d = torch.sort(dst, stable=True)[1]
            dst_keys = dst[perm_d]
        self.nbr_by_dst = (src if perm_d is None else src[perm_d]).to(torch.int32).contiguous()
        self.by_dst = EdgeOrder(None if perm_d is None else perm_d.to(torch.int32),
                                build_segment_items(_rowptr_from_sorted(dst_keys, self.num_nodes), chunk, ne))
        perm_s = torch.sort(src, stable=True)[1]
        self.nbr_by_src = dst[perm_s].to(torch.int32).contiguous()
        self.by_src = EdgeOrder(perm_s.to(torch.int32),
                                build_segment_items(_rowptr_from_sorted(src[perm_s], self.num_src_nodes), chunk, ne))

    def _build_native(self, src, dst, chunk, dst_sorted):
        """Both orderings from ONE C call (gv_graph_index_build): no torch sort / searchsorted / cumsum dispatches and no
        host synchronisation; lists sized by their upper bounds, -1 padded.  Same arrays as the torch formulation above."""
        E, nd, ns, dev = self.num_edges, self.num_nodes, self.num_src_nodes, self.device
        self.src32, self.dst32 = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
        ci_d, cf_d, slots_d = _index_caps(E, nd, chunk)
        ci_s, cf_s, slots_s = _index_caps(E, ns, chunk)
        (perm_d, nbr_d, rp_d, it_d, fx_d, perm_s, nbr_s, rp_s, it_s, fx_s) = _carve_i32(
            dev, [0 if dst_sorted else E, E, nd + 1, 4 * ci_d, 4 * cf_d, E, E, ns + 1, 4 * ci_s, 4 * cf_s])
        ws, ws_bytes = _index_workspace(dev, E, max(nd, ns))
        lib.call('gv_graph_index_build', ptr(self.src32), ptr(self.dst32), E, nd, ns, 1 if dst_sorted else 0, chunk,
                 None if dst_sorted else ptr(perm_d), ptr(nbr_d), ptr(rp_d), ptr(it_d), ci_d, ptr(fx_d), cf_d, ptr(perm_s),
                 ptr(nbr_s), ptr(rp_s), ptr(it_s), ci_s, ptr(fx_s), cf_s, ptr(ws), ws_bytes, lib.stream())
        self.nbr_by_dst, self.nbr_by_src = nbr_d, nbr_s
        self.by_dst = EdgeOrder(None if dst_sorted else perm_d,
                                SegmentItems(it_d.view(-1, 4), fx_d.view(-1, 4), ci_d, cf_d, slots_d, rp_d, chunk, exact=False))
        self.by_src = EdgeOrder(perm_s, SegmentItems(it_s.view(-1, 4), fx_s.view(-1, 4), ci_s, cf_s, slots_s, rp_s, chunk, exact=False))

    def lds_order(self, side: str, max_edges: int) -> 'LdsOrder':
        """Super-items of one ordering for the LDS-resident K1 kernel (csrc/k_lds.hip); built once per graph, cached."""
        key = (side, int(max_edges))
        hit = self._lds_seg_cache.get(key)
        if hit is None:
            own = self.by_dst.seg if side == 'dst' else self.by_src.seg
            hit = self._lds_seg_cache[key] = LdsOrder.build(own.rowptr, self.num_edges, int(max_edges))
        return hit

    def coef_in_src_order(self, coef: torch.Tensor) -> torch.Tensor:
        """Per-edge coefficients (given in the caller's edge order) permuted into the by-source order of the backward-x
        aggregation, so that launch reads them directly instead of through ``coef_idx`` (a dependent load per 64-edge
        batch).  Cached on the tensor's identity and version: the edge norm of a graph is the same every step."""
        key = (coef.data_ptr(), coef._version, coef.numel())
        hit = getattr(self, '_coef_src_cache', None)
        if hit is None or hit[0] != key:      # the entry holds the tensor, so its address cannot be recycled while cached
            hit = self._coef_src_cache = (key, coef.reshape(-1)[self.by_src.perm.long()].contiguous(), coef)
        return hit[1]

    def dst_chunks(self, n_chunks: int):
        """Cut the destination rows into ``n_chunks`` contiguous blocks of EQUAL ROW COUNT and return, per block,
        (row0, row1, SegmentItems restricted to those rows).  Used by the multi-GPU forward to start the all-reduce
        of one row block while the next one is still being aggregated.  The cut points depend on the node count only:
        every rank must slice the aggregate identically (its own edge block would give every rank different cuts and
        mismatched collectives).  Cached; synchronises once when built."""
        hit = self._chunk_cache.get(n_chunks)
        if hit is not None:
            return hit
        seg = self.by_dst.seg
        n_chunks = max(1, min(int(n_chunks), self.num_nodes))
        rows = sorted(set([0] + [(self.num_nodes * c) // n_chunks for c in range(1, n_chunks)] + [self.num_nodes]))
        # upper-bound-sized lists (sync-free build) end in -1 entries: cut them off first (this method synchronises anyway)
        n_items = int((seg.items[:seg.n_items, 0] >= 0).sum())
        n_fix = int((seg.fix[:seg.n_fix, 0] >= 0).sum()) if seg.n_fix > 0 else 0
        item_seg = seg.items[:n_items, 0].contiguous().to(torch.int64)
        fix_seg = seg.fix[:n_fix, 0].contiguous().to(torch.int64)
        bounds = torch.tensor(rows, device=seg.items.device, dtype=torch.int64)
        ib = torch.searchsorted(item_seg, bounds).tolist()
        fb = torch.searchsorted(fix_seg, bounds).tolist() if n_fix > 0 else [0] * len(rows)
        out = []
        for c in range(len(rows) - 1):
            sub = SegmentItems(seg.items[ib[c]:ib[c + 1]], seg.fix[fb[c]:fb[c + 1]] if n_fix > 0 else seg.fix,
                               ib[c + 1] - ib[c], fb[c + 1] - fb[c], seg.n_slots, seg.rowptr, seg.chunk)
            out.append((rows[c], rows[c + 1], sub))
        self._chunk_cache[n_chunks] = out
        return out

    def row_blocks(self, side: str, blocks):
        """Work items of the ``side`` ('dst' / 'src') ordering restricted to BLOCKS of rows: ``blocks`` = one list of (row0, row1)
        ranges per block (a block of the pipelined multi-GPU exchanges is the same slot-row range of every rank: ``world`` ranges of
        the table).  Returns one SegmentItems per block (item / fix-up lists = the concatenated slices of the ordering's lists;
        partial slots keep their numbers).  Cached per (side, blocks); synchronises once when built."""
        key = (side, tuple(tuple((int(a), int(b)) for a, b in blk) for blk in blocks))
        hit = self._chunk_cache.get(key)
        if hit is not None:
            return hit
        seg = (self.by_dst if side == 'dst' else self.by_src).seg
        n_items = int((seg.items[:seg.n_items, 0] >= 0).sum())
        n_fix = int((seg.fix[:seg.n_fix, 0] >= 0).sum()) if seg.n_fix > 0 else 0
        item_seg = seg.items[:n_items, 0].contiguous().to(torch.int64)
        fix_seg = seg.fix[:n_fix, 0].contiguous().to(torch.int64)
        flat = [r for blk in key[1] for ab in blk for r in ab]
        bounds = torch.tensor(flat, device=seg.items.device, dtype=torch.int64)
        ib = torch.searchsorted(item_seg, bounds).tolist()
        fb = torch.searchsorted(fix_seg, bounds).tolist() if n_fix > 0 else [0] * len(flat)
        out, q = [], 0
        for blk in key[1]:
            it, fx = [], []
            for _ in blk:
                it.append(seg.items[ib[q]:ib[q + 1]])
                if n_fix > 0:
                    fx.append(seg.fix[fb[q]:fb[q + 1]])
                q += 2
            items = torch.cat(it, 0).contiguous() if it else seg.items[:0]
            fix = torch.cat(fx, 0).contiguous() if fx else seg.fix[:0]
            if fix.shape[0] == 0:
                fix = seg.fix[:1] if seg.fix.shape[0] else torch.full((1, 4), -1, dtype=torch.int32, device=seg.items.device)
                nf = 0
            else:
                nf = int(fix.shape[0])
            if items.shape[0] == 0:
                items = torch.full((1, 4), -1, dtype=torch.int32, device=seg.items.device)
                ni = 0
            else:
                ni = int(items.shape[0])
            out.append(SegmentItems(items, fix, ni, nf, seg.n_slots, seg.rowptr, seg.chunk))
        self._chunk_cache[key] = out
        return out

    def relation_index(self, etypes: torch.Tensor, num_rels: int) -> 'RelationIndex':
        key = (etypes.data_ptr(), etypes._version, int(num_rels))
        hit = self._rel_cache.get(key)
        if hit is None:
            if len(self._rel_cache) > 8:
                self._rel_cache.clear()
            hit = self._rel_cache[key] = RelationIndex(self, etypes, num_rels)
        return hit


def xcd_order_items(seg: SegmentItems, key_by_pos: torch.Tensor, n_xcd: int = 8, group: int = 4):
    """Reorder a work-item list in place so that the workgroups ONE XCD receives (workgroups are dealt round-robin to the
    8 XCDs, `group` items per workgroup) are the items of one contiguous range of ``key_by_pos[item.begin]`` -- the rows
    an XCD gathers through that key then come from one window of the table and stay in its 4 MiB L2.  Items keep their
    contents (segment, range, slot), so results are bit-identical; -1 padding entries sort last and stay -1."""
    n = seg.n_items
    if n == 0 or key_by_pos.numel() == 0:
        return
    items = seg.items[:n]
    valid = items[:, 0] >= 0
    begin = items[:, 1].long().clamp(0, key_by_pos.numel() - 1)
    key = torch.where(valid, key_by_pos[begin].long(), torch.full_like(begin, torch.iinfo(torch.int64).max))
    order = torch.argsort(key, stable=True)
    per = -(-n // n_xcd)
    per = -(-per // group) * group
    pos = torch.arange(n, device=items.device)
    xcd, k = pos // per, pos % per
    if K1_ITEMS_LARGEST_FIRST:      # inside an XCD's window: by length too (measured neutral for the grad-W launches; kept for one rule)
        size = (items[order, 2] - items[order, 1]).long()
        order = order[torch.argsort(xcd * (1 << 20) - torch.where(valid[order], size, torch.zeros_like(size)), stable=True)]
    slot = (k // group) * (n_xcd * group) + xcd * group + (k % group)       # final index of the p-th item in key order
    out = torch.full((per * n_xcd, 4), -1, dtype=torch.int32, device=items.device)
    out[slot] = items[order]
    seg.items, seg.n_items = out, per * n_xcd


class RelationIndex:
    def __init__(self, g: GraphIndex, etypes: torch.Tensor, num_rels: int, chunk: Optional[int] = None):
        if etypes.numel() != g.num_edges:
            raise ValueError(f'etypes has {etypes.numel()} entries for {g.num_edges} edges')
        native = g.sync_free and NATIVE_INDEX
        # (the native builder takes int32 ids: a batch sampler's int32 relation ids skip the int64 round trip)
        et = etypes.reshape(-1) if (native and etypes.dtype == torch.int32) else etypes.reshape(-1).to(torch.int64)
        # (device-built graphs come with device-built relation ids: no host round trip to validate them)
        if g.num_edges and not g.sync_free and (int(et.min()) < 0 or int(et.max()) >= num_rels):
            raise ValueError(f'edge types must lie in [0, {num_rels})')
        self.num_rels = int(num_rels)
        self.keepalive = etypes
        self._rel_sorted = {}
        chunk = chunk_for(g.num_edges, DEFAULT_CHUNK_REL) if chunk is None else int(chunk)
        if g.sync_free and NATIVE_INDEX:
            E, dev = g.num_edges, g.device
            et32 = et.to(torch.int32).contiguous()
            ci, cf, slots = _index_caps(E, self.num_rels, chunk)
            (self.et_by_dst, self.et_by_src, perm_r, self.src_by_rel, self.dst_by_rel, rp, it, fx) = _carve_i32(
                dev, [E, E, E, E, E, self.num_rels + 1, 4 * ci, 4 * cf])
            ws, ws_bytes = _index_workspace(dev, E, self.num_rels)
            lib.call('gv_relation_index_build', ptr(g.src32), ptr(g.dst32), ptr(et32), ptr(g.by_dst.perm), ptr(g.by_src.perm),
                     E, self.num_rels, chunk, ptr(self.et_by_dst), ptr(self.et_by_src), ptr(perm_r), ptr(self.src_by_rel),
                     ptr(self.dst_by_rel), ptr(rp), ptr(it), ci, ptr(fx), cf, ptr(ws), ws_bytes, lib.stream())
            self.by_rel = EdgeOrder(perm_r, SegmentItems(it.view(-1, 4), fx.view(-1, 4), ci, cf, slots, rp, chunk, exact=False))
            return
        self.et_by_dst = (et if g.by_dst.perm is None else et[g.by_dst.perm.long()]).to(torch.int32).contiguous()
        self.et_by_src = et[g.by_src.perm.long()].to(torch.int32).contiguous()
        perm_r = torch.sort(et, stable=True)[1]
        self.src_by_rel = g.src32[perm_r].contiguous()
        self.dst_by_rel = g.dst32[perm_r].contiguous()
        self.by_rel = EdgeOrder(perm_r.to(torch.int32),
                                build_segment_items(_rowptr_from_sorted(et[perm_r], self.num_rels), chunk,
                                                    g.num_edges if g.sync_free else None))
        # measured on the FB15k-237-shaped graph: grad-W 79 -> 59 us (2x2) and 120 -> 85 us (2x4); per-batch graphs skip it
        if _os.environ.get('GV_GRADW_XCD', '1') == '1' and not g.sync_free and self.by_rel.seg.n_items >= 1024:
            self._xcd_order_items()

    def grouped_order(self, g: 'GraphIndex', side: str, n_groups: int = 8, chunk: int = DEFAULT_CHUNK):
        """Aggregation order for relation-weight tables that do not fit one XCD's L2 (h = 500: 4.7-9.5 MB against 4 MiB).

        The relation types are cut into ``n_groups`` ranges of ~equal edge counts and every row's edges are ordered by
        (row, group): a work item is one row's edges of ONE group (<= chunk of them), its partial row goes to a slot, and
        the existing fix-up pass adds a row's slots in group order.  The item list is laid out so that the workgroups one
        XCD receives (round-robin dealing) all belong to one group: that XCD's L2 then holds 1/n_groups of the weight table
        and the per-edge weight reads stop going to the Infinity Cache.  Kernels are unchanged (items / slots / fix-ups are
        their normal vocabulary); the sum over a row's edges is taken in (group, neighbour, relation) order instead of
        (neighbour, relation) order -- deterministic, equal to the plain order up to fp32 rounding.
        side 'dst': rows = destinations (forward); 'src': rows = sources (backward w.r.t. x).
        Returns (SegmentItems, nbr int32 [E], etype int32 [E], perm int64 [E] original edge id per position)."""
        cache = self.__dict__.setdefault('_grouped', {})
        hit = cache.get((side, n_groups, chunk))
        if hit is not None:
            return hit
        dev, E, G = g.device, g.num_edges, int(n_groups)
        et = self.keepalive.reshape(-1).to(torch.int64)
        rows, nbrs = (g.dst32, g.src32) if side == 'dst' else (g.src32, g.dst32)
        n_rows = g.num_nodes if side == 'dst' else g.num_src_nodes
        cnt = torch.bincount(et, minlength=self.num_rels)
        before = torch.cumsum(cnt, 0) - cnt
        grp_of_rel = torch.clamp(before * G // max(E, 1), max=G - 1)
        key = rows.long() * G + grp_of_rel[et]
        perm = torch.sort(key, stable=True)[1]                  # (row, group), then the caller's edge order
        key_s = key[perm]
        seg_key, seg_cnt = torch.unique_consecutive(key_s, return_counts=True)
        seg_start = torch.cumsum(seg_cnt, 0) - seg_cnt
        n_ch = (seg_cnt + chunk - 1) // chunk
        seg_of_item = torch.repeat_interleave(torch.arange(seg_key.numel(), device=dev), n_ch)
        k_in = torch.arange(seg_of_item.numel(), device=dev) - (torch.cumsum(n_ch, 0) - n_ch)[seg_of_item]
        begin = seg_start[seg_of_item] + k_in * chunk
        end = torch.minimum(begin + chunk, (seg_start + seg_cnt)[seg_of_item])
        item_row, item_grp = seg_key[seg_of_item] // G, seg_key[seg_of_item] % G
        has = torch.zeros(n_rows, dtype=torch.bool, device=dev)
        has[item_row] = True
        empty = torch.nonzero(~has).reshape(-1)                  # rows without edges still get their (empty) item
        item_row = torch.cat([item_row, empty])
        item_grp = torch.cat([item_grp, empty % G])
        begin = torch.cat([begin, torch.zeros_like(empty)])
        end = torch.cat([end, torch.zeros_like(empty)])
        order = torch.sort(item_row, stable=True)[1]            # a row's items adjacent, groups ascending
        item_row, item_grp, begin, end = item_row[order], item_grp[order], begin[order], end[order]
        per_row = torch.bincount(item_row, minlength=n_rows)
        multi = per_row > 1
        in_multi = multi[item_row]
        slot = torch.where(in_multi, torch.cumsum(in_multi.long(), 0) - 1, torch.full_like(item_row, -1))
        n_slots = int(in_multi.sum())
        first_slot = torch.cumsum(torch.where(multi, per_row, torch.zeros_like(per_row)), 0) - per_row
        fix_rows = torch.nonzero(multi).reshape(-1)
        fix = torch.stack([fix_rows, first_slot[fix_rows], per_row[fix_rows], torch.zeros_like(fix_rows)], 1).to(torch.int32)
        items = torch.stack([item_row, begin, end, slot], 1).to(torch.int32).contiguous()
        # XCD placement: blocks of 4 items are dealt round-robin to the 8 XCDs; give XCD x the items of group x (mod 8)
        n_items = int(items.shape[0])
        xcd = (item_grp % 8)
        order2 = torch.sort(xcd, stable=True)[1]
        per_x = torch.bincount(xcd, minlength=8)
        width = int(-(-int(per_x.max()) // 4) * 4) if n_items else 4
        rank_in_x = torch.arange(n_items, device=dev) - (torch.cumsum(per_x, 0) - per_x)[xcd[order2]]
        pos = (rank_in_x // 4) * 32 + xcd[order2] * 4 + rank_in_x % 4
        placed = torch.full((width * 8, 4), -1, dtype=torch.int32, device=dev)
        placed[pos] = items[order2]
        rowptr_rows = torch.zeros(n_rows + 1, dtype=torch.int32, device=dev)      # shape carrier (row count) for the launches
        seg = SegmentItems(placed, fix.contiguous() if fix.numel() else torch.full((1, 4), -1, dtype=torch.int32, device=dev),
                           width * 8, int(fix.shape[0]), max(n_slots, 1), rowptr_rows, chunk)
        hit = cache[(side, n_groups, chunk)] = (seg, nbrs[perm].to(torch.int32).contiguous(), et[perm].to(torch.int32).contiguous(),
                                                perm)
        return hit

    def phase_order(self, g: 'GraphIndex', side: str, num_bases: int, blk_in: int, blk_out: int) -> Optional['PhaseOrder']:
        """Tiles x relation phases x waves edge lists for the K1 phase kernel (csrc/k_phase.hip), built once per static
        graph, side ('dst': forward, 'src': backward w.r.t. x) and block shape; None when no phase kernel covers the shape."""
        cache = self.__dict__.setdefault('_phases', {})
        key = (side, int(num_bases), int(blk_in), int(blk_out), PHASE_LDS_BYTES, PHASE_ROWS, PHASE_THREADS, PHASE_BUFFERS, phase_stream_on(), PHASE_GREEDY)
        if key not in cache:
            cache[key] = PhaseOrder.build(self, g, side, num_bases, blk_in, blk_out)
        return cache[key]

    def grouped_coef(self, coef: torch.Tensor, side: str, perm: torch.Tensor) -> torch.Tensor:
        """Per-edge coefficients in a grouped order (cached per side on the tensor's identity and version)."""
        key = (coef.data_ptr(), coef._version, coef.numel())
        cache = self.__dict__.setdefault('_grouped_coef', {})
        hit = cache.get(side)
        if hit is None or hit[0] != key:
            hit = cache[side] = (key, coef.reshape(-1)[perm].contiguous(), coef)       # holds the key tensor alive
        return hit[1]

    def grouped_coef_src(self, coef, perm):
        return self.grouped_coef(coef, 'src', perm)

    def dense_plan(self, g: 'GraphIndex'):
        """Extras of the dense-weight (`basis`) path, built once per index: 64-row GEMM tiles that never cross a relation
        boundary, and for every position of the by-destination / by-source orders the position of the same edge in the
        by-relation order (where its message row lives)."""
        hit = getattr(self, '_dense_plan', None)
        if hit is None:
            dev, E = g.device, g.num_edges
            rp = self.by_rel.seg.rowptr.long()
            n_t = (rp[1:] - rp[:-1] + 63) // 64                                   # tiles per relation
            rel_of_tile = torch.repeat_interleave(torch.arange(self.num_rels, device=dev), n_t)
            first = torch.cumsum(n_t, 0) - n_t
            k_in_rel = torch.arange(rel_of_tile.numel(), device=dev) - first[rel_of_tile]
            row0 = rp[:-1][rel_of_tile] + 64 * k_in_rel
            row1 = torch.minimum(row0 + 64, rp[1:][rel_of_tile])
            tiles = torch.stack([row0, row1, rel_of_tile, torch.zeros_like(row0)], 1).to(torch.int32).contiguous()
            inv = torch.empty(E, dtype=torch.int64, device=dev)
            inv[self.by_rel.perm.long()] = torch.arange(E, device=dev)
            pos_by_dst = (inv if g.by_dst.perm is None else inv[g.by_dst.perm.long()]).to(torch.int32).contiguous()
            pos_by_src = inv[g.by_src.perm.long()].to(torch.int32).contiguous()
            zeros = torch.zeros(max(E, 1), dtype=torch.int32, device=dev)
            hit = self._dense_plan = (tiles, int(tiles.shape[0]), pos_by_dst, pos_by_src, zeros)
        return hit

    def rel_sorted(self, g: 'GraphIndex', side: str, coef: Optional[torch.Tensor] = None):
        """The ``side`` ('dst' / 'src') ordering of a STATIC graph with every row's edges sorted by relation (stable: the reference's
        order within a relation): (nbr, etype, edge ids, coef in that order or None).  Row pointers and work items are those of
        ``g.by_dst`` / ``g.by_src`` -- only positions inside a row move.  The per-row aggregation kernels keep a relation's weights
        in registers while consecutive edges share it, so each repeated (row, relation) pair saves a weight fetch (26 % of the
        edges of the FB15k-237-shaped graph; far more on real knowledge graphs, whose rows use few relations)."""
        hit = self._rel_sorted.get(side)
        if hit is None:
            order = g.by_dst if side == 'dst' else g.by_src
            nbr = g.nbr_by_dst if side == 'dst' else g.nbr_by_src
            et = self.et_by_dst if side == 'dst' else self.et_by_src
            rp = order.seg.rowptr.long()
            n_seg, E = rp.numel() - 1, g.num_edges
            deg = rp[1:] - rp[:-1]
            row = torch.repeat_interleave(torch.arange(n_seg, device=rp.device), deg, output_size=E)
            # (GV_K1_REL_RUNS_MIN_DEG = x: only rows of >= x * num_rels edges are re-sorted, the others keep their neighbour order;
            # measured at FB15k-237 size: sorting every row is best, 1.099 ms per step against 1.102 / 1.107 / 1.109 / 1.116 for x = 0.5 .. 4)
            min_deg = int(float(_os.environ.get('GV_K1_REL_RUNS_MIN_DEG', '0')) * self.num_rels)
            sub = torch.where(deg[row] >= min_deg, et.long(), torch.zeros((), dtype=torch.int64, device=rp.device))
            p2 = torch.sort(row * self.num_rels + sub, stable=True)[1]
            eid = order.perm.long() if order.perm is not None else torch.arange(E, device=rp.device)
            hit = self._rel_sorted[side] = (nbr[p2].contiguous(), et[p2].contiguous(), eid[p2].to(torch.int32).contiguous(), {})
        c = None
        if coef is not None:
            key = (coef.data_ptr(), coef._version, coef.numel())
            ent = hit[3].get('coef')
            if ent is None or ent[0] != key:
                ent = hit[3]['coef'] = (key, coef.reshape(-1)[hit[2].long()].contiguous(), coef)
            c = ent[1]
        return hit[0], hit[1], hit[2], c

    def coef_in_rel_order(self, coef: torch.Tensor) -> torch.Tensor:
        """Per-edge coefficients permuted into the by-relation order of the grad-W launch (cached like
        GraphIndex.coef_in_src_order)."""
        key = (coef.data_ptr(), coef._version, coef.numel())
        hit = getattr(self, '_coef_rel_cache', None)
        if hit is None or hit[0] != key:
            hit = self._coef_rel_cache = (key, coef.reshape(-1)[self.by_rel.perm.long()].contiguous(), coef)
        return hit[1]

    def _xcd_order_items(self):
        """Workgroups of one XCD cover one window of destination rows: within a relation the edges are in destination
        order, so an item gathers g[dst] rows from a narrow window, and items of similar windows then share an L2."""
        xcd_order_items(self.by_rel.seg, self.dst_by_rel)


# K1 by relation phases (csrc/k_phase.hip).  GV_K1_PHASES: '0' never, '1' whenever a phase kernel exists, 'auto' (default):
# static graphs, where measured faster than the per-row kernels (tools/phase_bench.py, tools/scale_check_phase.py):
#   * the gathered table is HBM scale (>= PHASE_MIN_TABLE_BYTES: the per-row kernels then also pull the relation weights
#     through L2 misses -- 1 M x 200 table, 2 000 relation types: 10.7 -> 7.7 ms per launch), or
#   * the 5x10 / 10x5 blocks of the reference's default width h = 500 (20 kB of weights per relation: 892 -> 515 us and
#     965 -> 692 us on the FB15k-237-shaped graph).
# At h = 200 on FB15k-237 (cache-resident tables) the per-row kernels stay ahead: the phase kernel executes ~2x the
# instructions per edge (82 vs 40, rocprofv3 SQ_INSTS_*) and is issue-bound there.
K1_PHASES = _os.environ.get('GV_K1_PHASES', 'auto')
PHASE_MIN_EDGES = 100_000
PHASE_MIN_TABLE_BYTES = 192 << 20
PHASE_LDS_BYTES = int(_os.environ.get('GV_PHASE_LDS', str(160 * 1024)))    # weight buffer(s) of one workgroup
PHASE_ROWS = int(_os.environ.get('GV_PHASE_ROWS', '0'))                     # rows per wave (0: the shape's default)
PHASE_THREADS = int(_os.environ.get('GV_PHASE_THREADS', '0'))                # workgroup threads (0: the shape's default -- 1 024, or 768 for the 5x10 blocks)
PHASE_GREEDY = _os.environ.get('GV_PHASE_GREEDY', '1') == '1'                   # waves of a tile chosen per item so that the phases end together (0: snake deal)
PHASE_BUFFERS = int(_os.environ.get('GV_PHASE_BUFFERS', '1'))               # 1: twice the relations per phase (measured faster); 2: staging overlaps compute


def phase_stream_on():
    """GV_PHASE_STREAM as the library reads it (per call: tests and probes flip it in os.environ)."""
    return _os.environ.get('GV_PHASE_STREAM', '1') != '0'


def use_phases(gidx, blk_in, blk_out, transpose_w, table_rows, table_cols):
    if K1_PHASES == '0' or gidx.sync_free or gidx.num_edges == 0:
        return False
    if K1_PHASES == '1':
        return True
    if gidx.num_edges < PHASE_MIN_EDGES:
        return False
    if int(table_rows) * int(table_cols) * 4 >= PHASE_MIN_TABLE_BYTES:
        return True
    # h = 500 (5-wide blocks): 10-20 kB of block weights per edge overflow an XCD's L2 -- staged per phase they win on every launch
    return (blk_in, blk_out, bool(transpose_w)) in ((5, 10, False), (10, 5, True), (5, 5, False), (5, 5, True))


@dataclass
class PhaseOrder:
    off: torch.Tensor          # int32 [n_tiles * nw * n_phases + 1]
    nbr: torch.Tensor          # int32 [E]
    meta: torch.Tensor         # int32 [E]
    perm: torch.Tensor         # int64 [E] original edge id of each position
    tile_items: torch.Tensor   # int32 [n_tiles, nw*K, 4]
    n_tiles: int
    fix: torch.Tensor
    n_fix: int
    n_slots: int
    n_rows: int
    rows_per_wave: int
    rels_per_phase: int
    n_phases: int
    threads: int
    buffers: int
    packed_floats: int
    transpose: bool

    @staticmethod
    def build(ridx: 'RelationIndex', g: 'GraphIndex', side: str, num_bases: int, blk_in: int, blk_out: int):
        trans = side == 'src'
        plan = (_ct.c_int32 * 7)()
        # the geometries of more than 8 rows per wave exist in the streamed kernel only: with it switched off ask for 1 024 threads
        threads_req = PHASE_THREADS if (PHASE_THREADS or phase_stream_on()) else 1024
        if not lib.load().gv_rgcn_bdd_phase_plan(int(num_bases), int(blk_in), int(blk_out), 1 if trans else 0, ridx.num_rels,
                                                 PHASE_LDS_BYTES, PHASE_BUFFERS, PHASE_ROWS, threads_req, _ct.addressof(plan)):
            return None
        _bpl, _parts, K, G, n_phases, packed_floats, threads = (int(v) for v in plan)
        order = g.by_dst if side == 'dst' else g.by_src
        nbr_sorted = g.nbr_by_dst if side == 'dst' else g.nbr_by_src
        et_sorted = (ridx.et_by_dst if side == 'dst' else ridx.et_by_src).long()
        n_rows = g.num_nodes if side == 'dst' else g.num_src_nodes
        dev, E = g.device, g.num_edges
        seg = order.seg
        items = seg.items[:seg.n_items]
        items = items[items[:, 0] >= 0].long()                   # drop the -1 padding of upper-bound-sized lists
        n_items = int(items.shape[0])
        nw = threads // 64
        T = nw * K
        n_tiles = max(1, -(-n_items // T))
        # deal the items to (tile, wave, slot) heaviest first, snake order: similar edge totals per tile and per wave
        size = items[:, 2] - items[:, 1]
        by_size = torch.sort(size, descending=True, stable=True)[1]
        j = torch.arange(n_items, device=dev)
        rnd, pos = j // n_tiles, j % n_tiles
        tile_s = torch.where(rnd % 2 == 0, pos, n_tiles - 1 - pos)          # tile of the j-th heaviest item
        s_in = rnd                                                          # arrival index inside the tile, < T
        rw, pw = s_in // nw, s_in % nw
        wave_s = torch.where(rw % 2 == 0, pw, nw - 1 - pw)
        k_s = rw
        tile_of_item = torch.empty(n_items, dtype=torch.long, device=dev)
        wave_of_item = torch.empty_like(tile_of_item)
        k_of_item = torch.empty_like(tile_of_item)
        tile_of_item[by_size], wave_of_item[by_size], k_of_item[by_size] = tile_s, wave_s, k_s
        if PHASE_GREEDY and n_items > 0 and E > 0:
            # The tile of an item stays (equal edge totals per tile); its WAVE is chosen so that the phases end together: a phase of a
            # tile ends with the wave that holds the longest list, and the snake deal balances a wave's TOTAL, not its 17-32 per-phase
            # lists (sum over tiles and phases of the longest list: 2.09x the mean at the 5x10 blocks of the FB15k-237-shaped graph, 1.47x
            # at 5x5).  Greedy, heaviest item first, all tiles at once: the item goes to the wave (with a free slot) that raises the
            # tile's sum of per-phase maxima least (1.78x / 1.19x).  Rows keep their own summation order: results are bit-identical.
            pos_e = torch.arange(E, device=dev)
            item_of_e = torch.searchsorted(items[:, 1].contiguous(), pos_e, right=True) - 1
            hist = torch.bincount(item_of_e * n_phases + et_sorted // G, minlength=n_items * n_phases).view(n_items, n_phases).to(torch.int32)
            load = torch.zeros(n_tiles, nw, n_phases, dtype=torch.int32, device=dev)
            used = torch.zeros(n_tiles, nw, dtype=torch.long, device=dev)
            for s in range(T):
                lo, hi = s * n_tiles, min((s + 1) * n_tiles, n_items)
                if lo >= n_items:
                    break
                it, tl = by_size[lo:hi], tile_s[lo:hi]               # round s: one item per tile
                h = hist[it]
                lt = load[tl]
                cand = lt + h[:, None, :]
                cost = torch.maximum(lt.max(1).values[:, None, :], cand).sum(2).double() + 1e-3 * cand.sum(2).double()
                cost = torch.where(used[tl] >= K, torch.full_like(cost, float('inf')), cost)
                w = cost.argmin(1)
                wave_of_item[it] = w
                k_of_item[it] = used[tl, w]
                load[tl, w] += h
                used[tl, w] += 1
        tile_items = torch.full((n_tiles, T, 4), -1, dtype=torch.int32, device=dev)
        flat = (tile_of_item * T + wave_of_item * K + k_of_item)
        ti = torch.zeros(n_items, 4, dtype=torch.int32, device=dev)
        ti[:, 0], ti[:, 1] = items[:, 0].to(torch.int32), items[:, 3].to(torch.int32)
        tile_items.view(-1, 4)[flat] = ti
        # edge position -> item (items are contiguous position ranges in row order; an empty item shares its begin with
        # the next one, which searchsorted(right) resolves to the later, non-empty item)
        pos_e = torch.arange(E, device=dev)
        item_of = torch.searchsorted(items[:, 1].contiguous(), pos_e, right=True) - 1
        phase = et_sorted // G
        key = (tile_of_item[item_of] * nw + wave_of_item[item_of]) * n_phases + phase
        # a wave's lists are contiguous (it streams its metadata); inside a (tile, wave, phase) list: by item slot (the kernel walks the slots with static accumulators), then in
        # the caller's edge order -> a fixed summation order per row
        perm_pos = torch.sort(key * K + k_of_item[item_of], stable=True)[1]
        counts = torch.bincount(key, minlength=n_tiles * nw * n_phases)
        off = torch.zeros(n_tiles * nw * n_phases + 1, dtype=torch.int32, device=dev)
        off[1:] = torch.cumsum(counts, 0).to(torch.int32)
        meta = (((et_sorted - phase * G) << 4) | k_of_item[item_of])[perm_pos].to(torch.int32).contiguous()
        nbr = nbr_sorted[perm_pos].contiguous()
        perm = perm_pos if order.perm is None else order.perm.long()[perm_pos]
        n_fix = int((seg.fix[:seg.n_fix, 0] >= 0).sum()) if seg.n_fix > 0 else 0
        return PhaseOrder(off, nbr, meta, perm, tile_items.contiguous(), n_tiles, seg.fix, n_fix, seg.n_slots, n_rows, K, G,
                          n_phases, threads, PHASE_BUFFERS, packed_floats, trans)

    def coef(self, coef: torch.Tensor) -> torch.Tensor:
        """Per-edge coefficients (caller's edge order) in list order; cached on the tensor (kept alive) and its version."""
        key = (coef.data_ptr(), coef._version, coef.numel())      # views / saved-tensor unpacks are new objects every call
        hit = getattr(self, '_coef', None)
        if hit is None or hit[0] != key:
            hit = self._coef = (key, coef, coef.reshape(-1)[self.perm].contiguous())      # holds the tensor: no address reuse
        return hit[2]


def pack_weight_phase(ph: PhaseOrder, weight, num_bases, blk_in, blk_out):
    """Lane-packed [parts][R][NQ][L] copy of a bdd relation-weight matrix for the phase kernel of one launch kind."""
    weight = _chk(weight, name='weight')
    packed = torch.empty(ph.packed_floats, dtype=torch.float32, device=weight.device)
    lib.call('gv_rgcn_bdd_pack_weight_phase', ptr(weight), weight.shape[0], num_bases, blk_in, blk_out,
             1 if ph.transpose else 0, ptr(packed), lib.stream())
    return packed


def bdd_aggregate_phases(ph: PhaseOrder, coef_p, feat, weight_packed, num_rels, num_bases, blk_in, blk_out, addend=None,
                         act=ACT_NONE, keep=None, keep_scale=1.0, out=None):
    """gv_rgcn_bdd_aggregate_phases: K1 with the relation weights staged through LDS phase by phase (``coef_p`` already in
    list order: PhaseOrder.coef; ``weight_packed``: pack_weight_phase)."""
    feat, ld_feat = _row_major(feat, 'feat')
    out_dim = num_bases * blk_out
    if feat.shape[1] != num_bases * blk_in:
        raise ValueError(f'feat has {feat.shape[1]} columns, expected num_bases*blk_in = {num_bases * blk_in}')
    if weight_packed.numel() != ph.packed_floats:
        raise ValueError('weight_packed does not have the size the phase plan asks for')
    if out is None:
        out = torch.empty(ph.n_rows, out_dim, dtype=torch.float32, device=feat.device)
    ld_add = 0
    if addend is not None:
        addend, ld_add = _row_major(addend, 'addend')
        if tuple(addend.shape) != (ph.n_rows, out_dim):
            raise ValueError('addend shape mismatch')
    if keep is not None:
        _chk(keep, torch.uint8, 'keep')
        if tuple(keep.shape) != (ph.n_rows, out_dim):
            raise ValueError('keep shape mismatch')
    if coef_p is not None:
        coef_p = _chk(coef_p.reshape(-1), name='coef')
    partial = torch.empty(ph.n_slots, out_dim, dtype=torch.float32, device=feat.device) if ph.n_fix > 0 else None
    ld_out = out.stride(0) if ph.n_rows > 1 else out_dim
    tag = f'agg_{"T" if ph.transpose else "N"}_{blk_in}x{blk_out}_nb{num_bases}'
    timed = lib.TIMER is not None
    lib.call('gv_rgcn_bdd_aggregate_phases', ptr(ph.off), ptr(ph.nbr), ptr(ph.meta), ptr(coef_p), ptr(ph.tile_items),
             ph.n_tiles, ptr(ph.fix), 0 if timed else ph.n_fix, ptr(feat), ld_feat, ptr(weight_packed), num_rels, num_bases,
             blk_in, blk_out, 1 if ph.transpose else 0, ph.rows_per_wave, ph.rels_per_phase, ph.buffers, ph.threads, ptr(addend), ld_add,
             act, ptr(keep), float(keep_scale), ptr(out), ld_out, ptr(partial), lib.stream(), tag=tag)
    if timed and ph.n_fix > 0:
        lib.call('gv_rgcn_bdd_fixup', ptr(ph.fix), ph.n_fix, ptr(partial), out_dim, ptr(addend), ld_add, act, ptr(keep),
                 float(keep_scale), ptr(out), ld_out, lib.stream())
    return out


@dataclass
class LdsOrder:
    """Work lists of the LDS-resident K1 kernel over one row ordering (rowptr): SUPER-ITEMS = runs of consecutive rows with
    <= G edges in all (one coalesced metadata fetch each) or <= G-edge slices of longer rows (partial slots + a fix-up
    entry per such row), the row of every edge position, and the rows without edges."""
    sitems: torch.Tensor       # int32 [n_sitems, 4] {first edge position, end position, partial slot (-1: whole rows), 0}
    n_sitems: int
    erow: torch.Tensor         # int32 [E]
    empty: torch.Tensor        # int32 [n_empty]
    n_empty: int
    fix: torch.Tensor          # int32 [n_fix, 4] {row, first slot, slices, 0}
    n_fix: int
    n_slots: int
    n_rows: int
    max_edges: int

    @staticmethod
    def build(rowptr: torch.Tensor, n_edges: int, G: int = 64) -> 'LdsOrder':
        """Small rows (<= G/2 edges) are grouped by the G/2-wide window their first edge falls in -- a group then spans at
        most G/2 - 1 + G/2 < G positions -- and never across a longer row; a longer row is its own item, cut into G-edge
        slices beyond G edges.  Torch ops on the device + one read-back of the list sizes (a static graph's index)."""
        dev = rowptr.device
        rp = rowptr.long()
        n_rows = rp.numel() - 1
        deg = rp[1:] - rp[:-1]
        rows = torch.arange(n_rows, device=dev)
        half = max(1, G // 2)
        small = (deg > 0) & (deg <= half)
        big = deg > half
        empty = rows[deg == 0].to(torch.int32).contiguous()
        erow = torch.repeat_interleave(rows, deg).to(torch.int32).contiguous()
        lists = []
        # small rows: group key = (long rows before it, window of its first edge)
        rs = rows[small]
        if rs.numel():
            nbig_before = torch.cumsum(big.long(), 0)[rs]
            key = nbig_before * (int(n_edges) // half + 2) + rp[rs] // half
            first = torch.ones_like(key, dtype=torch.bool)
            first[1:] = key[1:] != key[:-1]
            gi = torch.nonzero(first).flatten()                       # index into rs of every group's first row
            last = torch.cat([gi[1:] - 1, torch.tensor([rs.numel() - 1], device=dev)])
            e0, e1 = rp[rs[gi]], rp[rs[last] + 1]
            lists.append(torch.stack([e0, e1, torch.full_like(e0, -1), torch.zeros_like(e0)], 1))
        rb = rows[big]
        fix = torch.zeros(0, 4, dtype=torch.int32, device=dev)
        n_slots = 0
        if rb.numel():
            nsl = (deg[rb] + G - 1) // G
            split = nsl > 1
            slot0 = torch.cumsum(torch.where(split, nsl, torch.zeros_like(nsl)), 0) - torch.where(split, nsl, torch.zeros_like(nsl))
            n_slots = int(torch.where(split, nsl, torch.zeros_like(nsl)).sum())
            rep = torch.repeat_interleave(torch.arange(rb.numel(), device=dev), nsl)
            kk = torch.arange(rep.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(nsl, 0) - nsl, nsl)
            e0 = rp[rb][rep] + kk * G
            e1 = torch.minimum(e0 + G, rp[rb + 1][rep])
            slot = torch.where(split[rep], slot0[rep] + kk, torch.full_like(kk, -1))
            lists.append(torch.stack([e0, e1, slot, torch.zeros_like(e0)], 1))
            fr = rb[split]
            fix = torch.stack([fr, slot0[split], nsl[split], torch.zeros_like(fr)], 1).to(torch.int32).contiguous()
        if lists:
            sit = torch.cat(lists)
            sit = sit[torch.sort(sit[:, 0], stable=True)[1]].to(torch.int32).contiguous()
        else:
            sit = torch.zeros(0, 4, dtype=torch.int32, device=dev)
        pad = torch.zeros(1, 4, dtype=torch.int32, device=dev)
        return LdsOrder(sit if sit.numel() else pad, int(sit.shape[0]), erow if erow.numel() else torch.zeros(1, dtype=torch.int32, device=dev),
                        empty if empty.numel() else torch.zeros(1, dtype=torch.int32, device=dev), int(empty.numel()),
                        fix if fix.numel() else pad, int(fix.shape[0]), n_slots, n_rows, int(G))


K1_LDS = _os.environ.get('GV_K1_LDS', 'auto')               # LDS-resident relation weights: 'auto' | '0'
K1_LDS_WORKGROUPS = int(_os.environ.get('GV_K1_LDS_WGS', '0'))      # 0: one workgroup per CU
K1_LDS_G = int(_os.environ.get('GV_K1_LDS_G', '64'))                # most edges of a super-item (the kernel takes up to 64)
_LDS_PLANS = {}


LDS_MIN_EDGES = 100_000     # graphs indexed sync-free (rebuilt per step: mini-batches) stay on the per-row kernels below this size


def lds_graph(gidx, side=None, max_edges=None):
    """Does the LDS-resident kernel take this graph?  Its super-item lists are built with host read-backs (LdsOrder.build:
    data-dependent sizes) -- once per graph, ordering and list size, cached on the index.  A graph indexed sync-free takes the path
    from LDS_MIN_EDGES edges on (a large static graph whose lists are built during the eager warm-up steps); while the current
    stream is being CAPTURED a launch whose list (``side``, ``max_edges``) does not exist yet stays on the per-row kernels instead
    of aborting the capture with a synchronisation (a per-batch graph of that size inside graph_step.GraphedMiniBatchStep; a
    backward pass captured after an eager no-grad forward built only the 'dst' list) -- with a warning, once per graph, because
    the captured program then keeps the per-row kernels for good."""
    if gidx.sync_free and gidx.num_edges < LDS_MIN_EDGES:
        return False
    if torch.cuda.is_current_stream_capturing():
        have = bool(gidx._lds_seg_cache) if side is None else (side, int(max_edges)) in gidx._lds_seg_cache
        if not have:
            if not getattr(gidx, '_lds_capture_warned', False):
                gidx._lds_capture_warned = True
                import warnings
                warnings.warn('K1: the LDS-resident kernel\'s work list of this graph does not exist yet and cannot be built under '
                              'stream capture; the captured step keeps the per-row kernels (run one eager step first)')
            return False
    return True


def k1_bf16_applies(gidx, num_rels, num_bases, in_feat, out_feat):
    """True when a bdd layer of this shape on this graph runs its aggregations (forward and backward-x) on bf16 operands:
    --gemm-precision bf16 AND the LDS-resident kernel takes the layer (few relation types).  What an oracle has to mirror."""
    si, so = in_feat // num_bases, out_feat // num_bases
    return (k1_bf16() and lds_graph(gidx) and lds_plan(num_rels, num_bases, si, so, bf=True) is not None
            and lds_plan(num_rels, num_bases, so, si, bf=True) is not None)



def k1_bf16():
    """BASELINE configs[2]'s precision on K1: bf16 operands / fp32 accumulate where the LDS-resident kernel runs."""
    return _ops.GEMM_PRECISION == 'bf16'


def lds_plan(num_rels, num_bases, blk_in, blk_out, bf=None):
    """(column parts, floats of the packed table, most edges of a super-item, bf16 operands) when the LDS-resident K1 kernel
    exists for the block shape AND the relation table fits a CU's LDS (few relation types: WN18RR-shaped graphs), else None."""
    if K1_LDS == '0':
        return None
    bf = k1_bf16() if bf is None else bool(bf)
    key = (int(num_rels), int(num_bases), int(blk_in), int(blk_out), bf)
    if key not in _LDS_PLANS:
        plan = (_ct.c_int32 * 3)()
        ok = lib.load().gv_rgcn_bdd_lds_plan(key[1], key[2], key[3], key[0], 1 if bf else 0, _ct.addressof(plan))
        _LDS_PLANS[key] = (int(plan[0]), int(plan[1]), max(4, min(int(plan[2]), K1_LDS_G)), bf) if ok else None
    return _LDS_PLANS[key]


def pack_weight_lds(weight, num_bases, blk_in, blk_out, transpose_w, plan):
    weight = _chk(weight, name='weight')
    packed = torch.empty(plan[1], dtype=torch.float32, device=weight.device)
    lib.call('gv_rgcn_bdd_pack_weight_lds', ptr(weight), weight.shape[0], num_bases, blk_in, blk_out, 1 if transpose_w else 0,
             1 if plan[3] else 0, ptr(packed), lib.stream())
    return packed


def bdd_aggregate_lds(order: LdsOrder, nbr, etype, coef, coef_idx, feat, weight_packed, num_rels, num_bases, blk_in, blk_out,
                      transpose_w=False, addend=None, act=ACT_NONE, keep=None, keep_scale=1.0, out=None, plan=None):
    """gv_rgcn_bdd_aggregate_lds: K1 with every relation's block weights resident in LDS (``order``: GraphIndex.lds_order,
    ``weight_packed``: pack_weight_lds).  Same formula and epilogue as ``bdd_aggregate``; ``transpose_w`` only names the
    launch (the packing holds the orientation)."""
    feat, ld_feat = _row_major(feat, 'feat')
    n_seg = order.n_rows
    out_dim = num_bases * blk_out
    if feat.shape[1] != num_bases * blk_in:
        raise ValueError(f'feat has {feat.shape[1]} columns, expected num_bases*blk_in = {num_bases * blk_in}')
    plan = lds_plan(num_rels, num_bases, blk_in, blk_out) if plan is None else plan
    if plan is None or weight_packed.numel() != plan[1]:
        raise ValueError('weight_packed does not have the size the LDS plan asks for (or no plan for this shape)')
    if order.max_edges > plan[2]:
        raise ValueError(f'super-items of up to {order.max_edges} edges, the kernel takes {plan[2]}')
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
    partial = torch.empty(order.n_slots, out_dim, dtype=torch.float32, device=feat.device) if order.n_fix > 0 else None
    ld_out = out.stride(0) if n_seg > 1 else out_dim
    tag = f'agg_{"T" if transpose_w else "N"}_{blk_in}x{blk_out}_nb{num_bases}'
    timed = lib.TIMER is not None
    lib.call('gv_rgcn_bdd_aggregate_lds', ptr(order.sitems), order.n_sitems, ptr(order.erow), ptr(order.empty), order.n_empty,
             ptr(order.fix), 0 if timed else order.n_fix, ptr(nbr), ptr(etype), ptr(coef), ptr(coef_idx), ptr(feat), ld_feat,
             ptr(weight_packed), num_rels, num_bases, blk_in, blk_out, 1 if plan[3] else 0, ptr(addend), ld_add, act, ptr(keep),
             float(keep_scale), ptr(out), ld_out, ptr(partial), K1_LDS_WORKGROUPS, lib.stream(), tag=tag)
    if timed and order.n_fix > 0:
        lib.call('gv_rgcn_bdd_fixup', ptr(order.fix), order.n_fix, ptr(partial), out_dim, ptr(addend), ld_add, act, ptr(keep),
                 float(keep_scale), ptr(out), ld_out, lib.stream())
    return out


class TripletIndex:
    """Index of a (T,3) triplet batch for the DistMult backward (K1 with 1x1 blocks).

    incidence list: 2T entries (entity, other entity, relation, triplet id), sorted by entity
    relation list : T entries sorted by relation
    """

    def __init__(self, triplets: torch.Tensor, num_entities: int, num_rels: int, chunk: Optional[int] = None,
                 chunk_rel: Optional[int] = None, sync_free: bool = False, locality: Optional[bool] = None):
        if not triplets.is_cuda:
            raise RuntimeError('TripletIndex needs a CUDA tensor; there is no CPU fallback')
        t = triplets if triplets.dtype == torch.int32 else triplets.to(torch.int64)
        self.T = int(t.shape[0])
        chunk = chunk_for(2 * self.T) if chunk is None else int(chunk)
        chunk_rel = chunk_for(self.T, DEFAULT_CHUNK_REL) if chunk_rel is None else int(chunk_rel)
        self.num_entities, self.num_rels = int(num_entities), int(num_rels)
        if locality is None:       # extra sorts per index: for a batch that is used many times (not rebuilt every step)
            locality = not sync_free and self.num_entities * 800 >= (4 << 20)
        self.trip32 = t.to(torch.int32).contiguous()
        if sync_free and not locality and NATIVE_INDEX:
            T, dev, ne, nr = self.T, t.device, self.num_entities, self.num_rels
            ci_i, cf_i, slots_i = _index_caps(2 * T, ne, chunk)
            ci_r, cf_r, slots_r = _index_caps(T, nr, chunk_rel)
            (self.inc_other, self.inc_rel, self.inc_tid, rp_i, it_i, fx_i, self.rel_s, self.rel_o, self.rel_tid, rp_r, it_r,
             fx_r) = _carve_i32(dev, [2 * T, 2 * T, 2 * T, ne + 1, 4 * ci_i, 4 * cf_i, T, T, T, nr + 1, 4 * ci_r, 4 * cf_r])
            ws, ws_bytes = _index_workspace(dev, 2 * T, max(ne, nr))
            lib.call('gv_triplet_index_build', ptr(self.trip32), T, ne, nr, chunk, chunk_rel, ptr(self.inc_other),
                     ptr(self.inc_rel), ptr(self.inc_tid), ptr(rp_i), ptr(it_i), ci_i, ptr(fx_i), cf_i, ptr(self.rel_s),
                     ptr(self.rel_o), ptr(self.rel_tid), ptr(rp_r), ptr(it_r), ci_r, ptr(fx_r), cf_r, ptr(ws), ws_bytes,
                     lib.stream())
            self.inc = SegmentItems(it_i.view(-1, 4), fx_i.view(-1, 4), ci_i, cf_i, slots_i, rp_i, chunk, exact=False)
            self.rel = SegmentItems(it_r.view(-1, 4), fx_r.view(-1, 4), ci_r, cf_r, slots_r, rp_r, chunk_rel, exact=False)
            self.fwd_order = self.pos3 = None
            return
        t = t.to(torch.int64)
        s, r, o = t[:, 0], t[:, 1], t[:, 2]
        ent = torch.cat([s, o])
        other = torch.cat([o, s])
        rel2 = torch.cat([r, r])
        tid = torch.arange(self.T, device=t.device).repeat(2)
        perm = torch.sort(ent, stable=True)[1]
        self.inc_other = other[perm].to(torch.int32).contiguous()
        self.inc_rel = rel2[perm].to(torch.int32).contiguous()
        self.inc_tid = tid[perm].to(torch.int32).contiguous()
        if locality:
            inv_inc = torch.empty_like(perm)
            inv_inc[perm] = torch.arange(2 * self.T, device=t.device)
        self.inc = build_segment_items(_rowptr_from_sorted(ent[perm], self.num_entities), chunk,
                                       2 * self.T if sync_free else None)
        # DistMult forward walks the triplets in subject order (XCD windows of the embedding table, see k_distmult_bce);
        # the by-relation list of its weight gradient is ordered by (relation, subject) for the same reason
        self.fwd_order = torch.sort(s, stable=True)[1].to(torch.int32).contiguous() if locality else None
        perm_r = torch.sort(r * self.num_entities + s if locality else r, stable=True)[1]
        self.rel_s = s[perm_r].to(torch.int32).contiguous()
        self.rel_o = o[perm_r].to(torch.int32).contiguous()
        self.rel_tid = perm_r.to(torch.int32).contiguous()
        self.pos3 = None
        if locality:
            inv_rel = torch.empty_like(perm_r)
            inv_rel[perm_r] = torch.arange(self.T, device=t.device)
            # where triplet t sits in the two backward orders (gv_bce_grad scatters dL/dscore there: no coef_idx indirection)
            self.pos3 = torch.stack([inv_inc[:self.T], inv_inc[self.T:], inv_rel], dim=1).to(torch.int32).contiguous()
        self.rel = build_segment_items(_rowptr_from_sorted(r[perm_r], self.num_rels), chunk_rel,
                                       self.T if sync_free else None)
        if locality and self.T >= 65536:
            xcd_order_items(self.rel, self.rel_s)          # subject windows per XCD for the w_relation gradient
