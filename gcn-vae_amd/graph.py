"""Graph handle with the DGL-0.4 ``DGLGraph`` surface the reference touches
(kgvae/utils.py:127-150, kgvae/link_predict.py:95-100, :216) plus the device index cache the
HIP kernels need.  Edge arrays stay on the host until a layer asks for ``device_index``."""
import numpy as np
import torch

from . import ops


class _EdgeView:
    def __init__(self, g):
        src, dst = g.edges()
        self.src = {k: v[src.to(v.device)] for k, v in g.ndata.items()}
        self.dst = {k: v[dst.to(v.device)] for k, v in g.ndata.items()}
        self.data = g.edata


class KGraph:
    def __init__(self):
        self._n = 0
        self._dev_edges = None
        self._src = torch.zeros(0, dtype=torch.int64)
        self._dst = torch.zeros(0, dtype=torch.int64)
        self.ndata, self.edata = {}, {}
        self._index = {}

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def from_device_edges(cls, num_nodes, src, dst, dst_sorted=None):
        """Handle over edge lists that already live on a ROCm device (device_sampling.DeviceSampler): the
        kernels' index is built from them directly; a host copy is made only if ``edges()`` is asked for.
        ``dst_sorted=True``: the caller guarantees the reference's (dst, src, rel) edge order, which saves the check."""
        g = cls()
        g._dst_sorted = dst_sorted
        g._n = int(num_nodes)
        g._dev_edges = (src, dst)          # int32 or int64, as the builder made them: no conversion kernel per batch
        g._src = g._dst = None
        return g

    def _host_edges(self):
        if self._src is None:
            self._src, self._dst = (t.cpu().to(torch.int64) for t in self._dev_edges)
        return self._src, self._dst

    def add_nodes(self, n):
        self._n += int(n)
        self._index.clear()

    def add_edges(self, src, dst):
        self._host_edges() if getattr(self, '_dev_edges', None) is not None else None
        self._dev_edges = None
        s = torch.as_tensor(np.asarray(src) if not isinstance(src, torch.Tensor) else src, dtype=torch.int64).cpu()
        d = torch.as_tensor(np.asarray(dst) if not isinstance(dst, torch.Tensor) else dst, dtype=torch.int64).cpu()
        if s.shape != d.shape:
            raise ValueError('src and dst must have the same length')
        if s.numel() and (int(torch.max(torch.max(s), torch.max(d))) >= self._n or int(torch.min(torch.min(s), torch.min(d))) < 0):
            raise ValueError('edge endpoint out of range; call add_nodes first')
        self._src = torch.cat([self._src, s.reshape(-1)])
        self._dst = torch.cat([self._dst, d.reshape(-1)])
        self._index.clear()

    # -- queries --------------------------------------------------------------------------------
    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        if getattr(self, '_dev_edges', None) is not None:
            return int(self._dev_edges[0].numel())
        return int(self._src.numel())

    def __len__(self):
        return self._n

    def edges(self):
        if getattr(self, '_dev_edges', None) is not None:
            return self._host_edges()
        return self._src, self._dst

    def in_degrees(self, nodes=None):
        deg = torch.bincount(self.edges()[1], minlength=self._n)
        if nodes is None:
            return deg
        return deg[torch.as_tensor(list(nodes) if not isinstance(nodes, torch.Tensor) else nodes, dtype=torch.int64)]

    def local_var(self):
        g = KGraph()
        g._n, g._src, g._dst = self._n, self._src, self._dst
        g._dev_edges = getattr(self, '_dev_edges', None)
        g.ndata, g.edata = dict(self.ndata), dict(self.edata)
        g._index = self._index          # the device index depends on (src, dst) only
        return g

    def apply_edges(self, fn):
        self.edata.update(fn(_EdgeView(self)))

    # -- device side ----------------------------------------------------------------------------
    def device_index(self, device) -> 'ops.GraphIndex':
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('the R-GCN kernels run on a ROCm device only; there is no CPU fallback '
                               f'(asked for an index on {device})')
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        idx = self._index.get(key)
        if idx is None:
            if getattr(self, '_dev_edges', None) is not None:
                src, dst = (t.to(device) for t in self._dev_edges)
            else:
                src, dst = self._src.to(device), self._dst.to(device)
            # graphs of mini-batch size are rebuilt every step: build their index without host synchronisation
            small = src.numel() <= SYNC_FREE_MAX_EDGES
            idx = self._index[key] = ops.GraphIndex(src, dst, self._n, dst_sorted=getattr(self, '_dst_sorted', None),
                                                    sync_free=small)
        return idx


SYNC_FREE_MAX_EDGES = 200_000     # above this a graph is assumed to be built once: exact work-item lists

DGLGraph = KGraph


def graph_index_of(g, device):
    """``device_index`` of our handle, or an index built from any object exposing edges()/number_of_nodes()."""
    if hasattr(g, 'device_index'):
        return g.device_index(device)
    cache = g.__dict__.setdefault('_gv_index', {})
    key = str(device)
    if key not in cache:
        src, dst = g.edges()
        cache[key] = ops.GraphIndex(torch.as_tensor(src).to(device), torch.as_tensor(dst).to(device),
                                    g.number_of_nodes())
    return cache[key]
