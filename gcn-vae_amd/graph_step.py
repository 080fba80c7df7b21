"""The reference's real training mode -- one sampled sub-graph per step (kgvae/link_predict.py:200-236) -- as ONE hipGraph.

Everything between two optimiser updates is recorded once and replayed: the device batch sampler (edge sample -> relabel ->
negatives -> split -> (dst, src, rel)-ordered graph), the CSR / relation / triplet index builders, embedding gather, both
R-GCN layers, reparameterisation, the loss head, backward, clip + Adam.  What makes that possible:

  * static shapes: node arrays are padded to cap = min(2 * sample_size, num_nodes) rows; how many of them exist stays in device
    memory (``DeviceBatch.rows_dev``) and the loss kernels read it there (include/gcnvae.h, rows_dev) -- padding rows are isolated
    nodes: they cost arithmetic, get zero gradient and take no part in any mean;
  * work-item lists sized by upper bounds and -1 padded (the index builders' sync-free form);
  * random draws keyed by counters that live in device memory and are advanced inside the graph (sampler batch number,
    dropout / noise tick), so every replay draws fresh numbers -- the same ones an eager run of the same seeds draws.

Eager launching costs ~2.3 ms per step here against ~0.8 ms of kernel time (the step is ~130 launches of 3-40 us); the replay
removes the host from the loop.  tests/test_gpu_model.py holds the replayed steps equal to eager ones.
"""
import os as _os

import torch

from . import ops
from .optim import FlatAdam


class _Rows:
    """What LinkPredict.triplet_index reads of the embedding it is handed: the row count and the device."""
    def __init__(self, n, device):
        self.shape, self.device = (int(n),), device


class GraphedMiniBatchStep:
    def __init__(self, model, optimizer, sampler, sample_size, split_size=0.5, negative_rate=10, num_mmd_rows=200):
        if not isinstance(optimizer, FlatAdam):
            raise TypeError('the captured step needs FlatAdam (static gradient arena, two-launch clip + Adam)')
        if getattr(model, 'kl_param', 1) <= 0:
            raise ValueError('the captured mini-batch step needs kl_param > 0 (the loss head reads the device row count there)')
        self.model, self.opt, self.sampler = model, optimizer, sampler
        self.args = (int(sample_size), float(split_size), int(negative_rate))
        dev = sampler.device
        enc = model.encoder
        self.pick = None
        if getattr(model, 'mmd_param', 0) > 0 and hasattr(enc, 'mmd_index_override'):
            self.pick = torch.zeros(num_mmd_rows, dtype=torch.int64, device=dev)      # refilled on the device inside the step
        self.one = torch.ones((), device=dev)
        self.graph = None
        self.out = None
        self.side = torch.cuda.Stream(device=dev)
        # the triplet index (DistMult backward: ~12 launches of ~5 us, needed by the loss head) is built on its own stream beside
        # the encoder's forward instead of between the encoder and the loss head
        self.idx_side = torch.cuda.Stream(device=dev) if _os.environ.get('GV_MB_INDEX_SIDE', '1') == '1' else None

    def body(self):
        """The step, launched eagerly (also what the capture records).  The model's static-batch settings (device row count,
        device-side MMD pick, KL kept in the loss head) hold for the duration of the step only: an evaluation forward on
        another graph between two steps sees the model as it was (the recorded kernels keep what they captured)."""
        m, enc = self.model, self.model.encoder
        b = self.sampler.sample_static(*self.args, mmd_pick=self.pick)
        saved = (getattr(m, 'rows_dev', None), getattr(enc, 'rows_dev', None), getattr(enc, 'mmd_index_override', None),
                 getattr(enc, 'fuse_kl_with_reparam', None))
        m.rows_dev = b.rows_dev
        if hasattr(enc, 'rows_dev'):
            enc.rows_dev = b.rows_dev                      # flow_log_prob is a mean over the rows that exist
        if self.pick is not None:
            enc.mmd_index_override = self.pick
        if hasattr(enc, 'fuse_kl_with_reparam'):
            enc.fuse_kl_with_reparam = False               # the KL pass needs the device row count: it stays in the loss head here
        try:
            self.opt.zero_grad()
            batched = self._batch_indices(b)
            if not batched and self.idx_side is not None and hasattr(m, 'triplet_index'):
                # (the graph's own index the same way, on a second stream, waited for behind layer 1's self-loop product: 1.07 ms
                # against 0.94 -- the builders' small launches then compete with the forward pass's for the dispatcher)
                self.idx_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.idx_side):
                    m.triplet_index(_Rows(b.node_id.shape[0], b.samples.device), b.samples)      # cached: get_loss finds it built
            with ops.live_rows(b.rows_dev, b.node_id.shape[0]):      # the dense products skip the padding rows
                embed = m(b.g, b.node_id, b.edge_type, b.edge_norm)
                if not batched and self.idx_side is not None:
                    torch.cuda.current_stream().wait_stream(self.idx_side)
                loss, pred, kl, mmd = m.get_loss(b.g, embed, b.samples, b.labels)
                loss.backward(gradient=self.one.expand_as(loss))
            self.opt.step()
        finally:
            m.rows_dev = saved[0]
            if hasattr(enc, 'rows_dev'):
                enc.rows_dev = saved[1]
            if self.pick is not None:
                enc.mmd_index_override = saved[2]
            if saved[3] is not None:
                enc.fuse_kl_with_reparam = saved[3]
        self.batch = b
        return loss, pred, kl, mmd

    def _batch_indices(self, b):
        """All of the batch's orderings (graph by destination / source, relation, triplet incidences, triplets by relation) in
        shared launches (ops.build_batch_indices: 7 launches instead of ~30 dependent ones of ~4 us), handed to the places the
        modules look their indices up: the graph's per-device cache, the graph index's relation cache, the model's triplet cache.
        False = not applicable (GV_INDEX_BATCH=0, a model without those caches): the modules build their indices as they go."""
        m = self.model
        layers = [l for l in m.modules() if hasattr(l, 'num_rels') and hasattr(l, 'num_bases')]
        dev_edges = getattr(b.g, '_dev_edges', None)
        if not (ops.batch_index.BATCH_INDEX and ops.indices.NATIVE_INDEX and layers and dev_edges is not None and hasattr(m, 'w_relation')
                and hasattr(m, '_tidx_key') and b.edge_type.dtype == torch.int32):
            return False
        num_rels = {int(l.num_rels) for l in layers}
        if len(num_rels) != 1:
            return False
        src, dst = dev_edges
        dev = src.device
        n = b.node_id.shape[0]
        gidx, _, tidx = ops.build_batch_indices(src, dst, b.edge_type, n, num_rels.pop(), b.samples, n, m.w_relation.shape[0],
                                                dst_sorted=bool(getattr(b.g, '_dst_sorted', False)))
        b.g._index[(dev.type, dev.index if dev.index is not None else torch.cuda.current_device())] = gidx
        m._tidx, m._tidx_keepalive = tidx, b.samples
        m._tidx_key = (b.samples.data_ptr(), b.samples._version, tuple(b.samples.shape), n)
        return True

    def capture(self, warmup=3):
        """Eager warm-up steps (they are real training steps) on the capture stream, then the recording."""
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            for _ in range(warmup):
                self.body()
        torch.cuda.current_stream().wait_stream(self.side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.side):
            self.out = self.body()
        return self

    def eager_step(self):
        """One step launched eagerly ON THE CAPTURE STREAM (autograd pins a parameter's AccumulateGrad node to the stream of its
        first use; warming up elsewhere would drag that stream into the later capture)."""
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            out = self.body()
        torch.cuda.current_stream().wait_stream(self.side)
        return out

    def __call__(self):
        """One training step; returns (loss, predict_loss, kl, mmd) as device scalars (static storage: read before the next call)."""
        if self.graph is None:
            return self.eager_step()
        self.graph.replay()
        return self.out
