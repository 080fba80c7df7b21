"""All orderings of ONE sampled batch from shared launches (gv_triplet_lists + gv_build_csr_batch): the captured mini-batch step's index
builder (graph_step.GraphedMiniBatchStep).  The index classes themselves, and the builders that make one ordering at a time, live in
indices.py."""
import ctypes as _ct
import os as _os

import torch

from . import lib
from .indices import (DEFAULT_CHUNK_REL, EdgeOrder, GraphIndex, RelationIndex, SegmentItems, TripletIndex, _carve_i32, _index_caps,
                      chunk_for)
from .lib import ptr


class _CsrJob(_ct.Structure):
    """ctypes mirror of gv_csr_job (include/gcnvae.h; layout checked by tests/test_abi.py)."""
    _fields_ = [('keys', _ct.c_void_p), ('n', _ct.c_int64), ('n_seg', _ct.c_int32), ('chunk', _ct.c_int32),
                ('perm', _ct.c_void_p), ('rowptr', _ct.c_void_p), ('items', _ct.c_void_p), ('fix', _ct.c_void_p),
                ('items_cap', _ct.c_int32), ('fix_cap', _ct.c_int32), ('carry_src', _ct.c_void_p * 3),
                ('carry_out', _ct.c_void_p * 3)]


def _csr_job(keys, n_seg, chunk, perm, rowptr, items, ci, fix, cf, carries):
    j = _CsrJob()
    j.keys, j.n, j.n_seg, j.chunk = ptr(keys), int(keys.numel()), int(n_seg), int(chunk)
    j.perm, j.rowptr, j.items, j.fix, j.items_cap, j.fix_cap = ptr(perm), ptr(rowptr), ptr(items), ptr(fix), int(ci), int(cf)
    for k, (a, b) in enumerate(carries):
        j.carry_src[k], j.carry_out[k] = ptr(a), ptr(b)
    return j


BATCH_INDEX = _os.environ.get('GV_INDEX_BATCH', '1') == '1'


def build_batch_indices(src, dst, etypes, num_nodes, num_rels, triplets=None, num_entities=None, num_trip_rels=None,
                        dst_sorted=True):
    """The graph, relation and (optionally) triplet index of ONE sampled batch from two C calls -- gv_triplet_lists and
    gv_build_csr_batch: the five orderings (edges by destination / source / relation, triplet incidences by entity, triplets by
    relation) share their launches (six instead of ~30).  Same arrays as GraphIndex(sync_free=True), RelationIndex and
    TripletIndex(sync_free=True) build one after the other (tests/test_gpu_ops.py).  src / dst / etypes: int32 device tensors;
    triplets: (T, 3) int32 or int64.  Returns (GraphIndex, RelationIndex, TripletIndex or None)."""
    dev, E = src.device, int(src.numel())
    nd = ns = int(num_nodes)
    src32, dst32 = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
    et32 = etypes.reshape(-1).to(torch.int32).contiguous()
    chunk = chunk_for(E)
    chunk_r = chunk_for(E, DEFAULT_CHUNK_REL)
    ci_d, cf_d, slots_d = _index_caps(E, nd, chunk)
    ci_s, cf_s, slots_s = _index_caps(E, ns, chunk)
    ci_r, cf_r, slots_r = _index_caps(E, int(num_rels), chunk_r)
    (perm_d, nbr_d, rp_d, it_d, fx_d, perm_s, nbr_s, rp_s, it_s, fx_s, et_d, et_s, perm_r, src_r, dst_r, rp_r, it_r, fx_r) = _carve_i32(
        dev, [0 if dst_sorted else E, E, nd + 1, 4 * ci_d, 4 * cf_d, E, E, ns + 1, 4 * ci_s, 4 * cf_s,
              E, E, E, E, E, int(num_rels) + 1, 4 * ci_r, 4 * cf_r])
    jobs = [_csr_job(dst32, nd, chunk, None if dst_sorted else perm_d, rp_d, it_d, ci_d, fx_d, cf_d, [(src32, nbr_d), (et32, et_d)]),
            _csr_job(src32, ns, chunk, perm_s, rp_s, it_s, ci_s, fx_s, cf_s, [(dst32, nbr_s), (et32, et_s)]),
            _csr_job(et32, num_rels, chunk_r, perm_r, rp_r, it_r, ci_r, fx_r, cf_r, [(src32, src_r), (dst32, dst_r)])]
    keep = [src32, dst32, et32]
    tidx = None
    if triplets is not None:
        T, ne, nr = int(triplets.shape[0]), int(num_entities), int(num_trip_rels)
        trip_in = triplets.contiguous()
        if trip_in.dtype not in (torch.int32, torch.int64):
            raise TypeError('triplets: int32 or int64')
        chunk_t, chunk_tr = chunk_for(2 * T), chunk_for(T, DEFAULT_CHUNK_REL)
        ci_i, cf_i, slots_i = _index_caps(2 * T, ne, chunk_t)
        ci_q, cf_q, slots_q = _index_caps(T, nr, chunk_tr)
        (ent, other, rel2, tid, perm_i, cs, cr, co, inc_other, inc_rel, inc_tid, rp_i, it_i, fx_i, rel_s, rel_o, rel_tid, rp_q, it_q,
         fx_q, trip32) = _carve_i32(dev, [2 * T, 2 * T, 2 * T, 2 * T, 2 * T, T, T, T, 2 * T, 2 * T, 2 * T, ne + 1, 4 * ci_i, 4 * cf_i,
                                          T, T, T, nr + 1, 4 * ci_q, 4 * cf_q, 0 if trip_in.dtype == torch.int32 else 3 * T])
        trip32 = trip_in if trip_in.dtype == torch.int32 else trip32.view(T, 3)      # (int64 as the sampler returns it: narrowed in the same launch)
        lib.call('gv_triplet_lists', ptr(trip_in), 1 if trip_in.dtype == torch.int64 else 0, T, ptr(ent), ptr(other), ptr(rel2), ptr(tid),
                 ptr(cs), ptr(cr), ptr(co), None if trip_in.dtype == torch.int32 else ptr(trip32), lib.stream())
        jobs.append(_csr_job(ent, ne, chunk_t, perm_i, rp_i, it_i, ci_i, fx_i, cf_i, [(other, inc_other), (rel2, inc_rel), (tid, inc_tid)]))
        jobs.append(_csr_job(cr, nr, chunk_tr, rel_tid, rp_q, it_q, ci_q, fx_q, cf_q, [(cs, rel_s), (co, rel_o)]))
        keep += [trip_in, trip32, ent, other, rel2, tid, perm_i, cs, cr, co]
    arr = (_CsrJob * len(jobs))(*jobs)
    nbytes = int(lib.load().gv_build_csr_batch_workspace_bytes(_ct.addressof(arr), len(jobs)))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    lib.call('gv_build_csr_batch', _ct.addressof(arr), len(jobs), ptr(ws), nbytes, lib.stream())
    g = GraphIndex.__new__(GraphIndex)
    g.num_nodes, g.num_edges, g.num_src_nodes, g.device, g.sync_free = nd, E, ns, dev, True
    g._rel_cache, g._chunk_cache, g._lds_seg_cache = {}, {}, {}
    g.src32, g.dst32, g.nbr_by_dst, g.nbr_by_src = src32, dst32, nbr_d, nbr_s
    g.by_dst = EdgeOrder(None if dst_sorted else perm_d, SegmentItems(it_d.view(-1, 4), fx_d.view(-1, 4), ci_d, cf_d, slots_d, rp_d, chunk, exact=False))
    g.by_src = EdgeOrder(perm_s, SegmentItems(it_s.view(-1, 4), fx_s.view(-1, 4), ci_s, cf_s, slots_s, rp_s, chunk, exact=False))
    g._batch_keepalive = (keep, ws)
    r = RelationIndex.__new__(RelationIndex)
    r.num_rels, r.keepalive, r._rel_sorted = int(num_rels), etypes, {}
    r.et_by_dst, r.et_by_src, r.src_by_rel, r.dst_by_rel = et_d, et_s, src_r, dst_r
    r.by_rel = EdgeOrder(perm_r, SegmentItems(it_r.view(-1, 4), fx_r.view(-1, 4), ci_r, cf_r, slots_r, rp_r, chunk_r, exact=False))
    g._rel_cache[(etypes.data_ptr(), etypes._version, int(num_rels))] = r
    if triplets is not None:
        tidx = TripletIndex.__new__(TripletIndex)
        tidx.T, tidx.num_entities, tidx.num_rels, tidx.trip32 = T, ne, nr, trip32
        tidx.inc_other, tidx.inc_rel, tidx.inc_tid, tidx.rel_s, tidx.rel_o, tidx.rel_tid = inc_other, inc_rel, inc_tid, rel_s, rel_o, rel_tid
        tidx.inc = SegmentItems(it_i.view(-1, 4), fx_i.view(-1, 4), ci_i, cf_i, slots_i, rp_i, chunk_t, exact=False)
        tidx.rel = SegmentItems(it_q.view(-1, 4), fx_q.view(-1, 4), ci_q, cf_q, slots_q, rp_q, chunk_tr, exact=False)
        tidx.fwd_order = tidx.pos3 = None
    return g, r, tidx
