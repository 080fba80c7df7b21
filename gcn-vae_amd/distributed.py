"""Multi-GPU layer: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards by EDGE BLOCK (north_star): every rank holds the replicated parameters, its own
block of the edge list (CSR built locally) and its own slice of the decoder's triplets.

  forward   partial aggregate  agg_p = sum over the rank's edges     (K1, local)
            all-reduce(sum) of (N, out) node embeddings  || self-loop GEMM   <- the one exchange per layer
            epilogue (self loop + bias + activation + dropout), decoder on the rank's triplets
  backward  the gradient of the aggregate is all-reduced(sum) the same way (every rank's loss sees
            every rank's edges through the reduced embeddings) || bias / loop-weight gradients and the
            loop GEMM of the layer, then K1^T / grad-W run locally
  step      parameter gradients are averaged over ranks (one flat all-reduce), so the update equals
            the single-process gradient of  (1/P) sum_p loss_p  on the union graph.

``AllReduceSum`` is the same exchange as a device-agnostic autograd Function; the world_size-2 gloo
tests drive it (and ``shard_edges_by_relation`` / ``average_gradients``) on CPU tensors.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


class _Done:
    def wait(self):
        return None


def make_reduce_hook(group=None, async_op=True):
    """Callable for ``RelGraphConv.reduce_hook``: starts an in-place sum of ``t`` over the edge shards and
    returns a handle whose ``wait()`` orders the current stream behind it.  With ``async_op`` the collective
    runs on RCCL's own stream, so whatever the caller enqueues between the call and ``wait()`` (the layer's
    self-loop GEMM in forward; bias / loop-weight gradients and the loop GEMM in backward) overlaps with it."""
    def hook(t):
        if async_op:
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return _Done()
    return hook


class AllReduceSum(torch.autograd.Function):
    """y = sum_p x_p on every rank; backward: grad_x_p = sum_q grad_y_q."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def average_gradients(params, group=None):
    """One flat all-reduce over every parameter gradient, divided by the world size."""
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or world == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def average_flat(flat_grads, group=None):
    """The same for a FlatAdam gradient arena: one all-reduce, no packing."""
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        flat_grads.div_(world)


def shard_edges_by_relation(etypes, num_rels, world, rank):
    """Edge ids of ``rank``'s block: relations are cut into ``world`` contiguous ranges holding
    ~E/world edges each (whole relations, so each rank touches a disjoint slice of ``weight``)."""
    et = np.asarray(etypes)
    counts = np.bincount(et, minlength=num_rels)
    csum = np.cumsum(counts)
    total = csum[-1] if len(csum) else 0
    bounds = [0]
    for p in range(1, world):
        bounds.append(int(np.searchsorted(csum, total * p / world, side='left')) + 1 if total else 0)
    bounds.append(num_rels)
    bounds = np.minimum.accumulate(np.array(bounds[::-1]))[::-1]        # keep the cut points monotone
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    return np.nonzero((et >= lo) & (et < hi))[0], (lo, hi)


def global_in_degree_norm(dst_local, num_nodes, group=None, device=None):
    """1/in-degree over the UNION graph (every rank contributes its block's destinations)."""
    deg = torch.bincount(torch.as_tensor(dst_local, dtype=torch.int64), minlength=num_nodes).to(torch.float32)
    if device is not None:
        deg = deg.to(device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(deg, op=dist.ReduceOp.SUM, group=group)
    norm = torch.zeros_like(deg)
    nz = deg > 0
    norm[nz] = 1.0 / deg[nz]
    return norm
