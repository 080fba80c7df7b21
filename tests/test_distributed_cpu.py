"""world_size-2 gloo tests (CPU) of the multi-GPU scheme in gcn-vae_amd/distributed.py.

The exchange pattern (edge-block sharding by relation, all-reduce(sum) of partial node aggregates in
forward, all-reduce(sum) of their gradient in backward, one averaged all-reduce of parameter gradients)
is device independent; here the local arithmetic is the CPU oracle and the result must equal a
single-process run on the union graph with loss = mean of the ranks' losses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kgvae as okg
from oracle import rgcn as orgcn

N, R, H, NB, E, T = 60, 8, 8, 4, 500, 120


def make_problem():
    rs = np.random.RandomState(0)
    src, dst, et = rs.randint(0, N, E), rs.randint(0, N, E), rs.randint(0, R, E)
    trip = np.stack([rs.randint(0, N, T), rs.randint(0, R // 2, T), rs.randint(0, N, T)], 1)
    labels = (rs.rand(T) > 0.5).astype(np.float32)
    gen = torch.Generator().manual_seed(0)
    params = {
        'emb': torch.randn(N, H, generator=gen),
        'w1': torch.randn(R, NB * (H // NB) * (H // NB), generator=gen) * 0.5, 'b1': torch.randn(H, generator=gen) * 0.1,
        'l1': torch.randn(H, H, generator=gen) * 0.3,
        'w2': torch.randn(R, NB * (H // NB) * (H // NB), generator=gen) * 0.5, 'b2': torch.randn(H, generator=gen) * 0.1,
        'l2': torch.randn(H, H, generator=gen) * 0.3,
        'w_rel': torch.randn(R // 2, H, generator=gen) * 0.5,
    }
    return src, dst, et, trip, labels, params


def layer(x, src, dst, et, norm, w, b, lw, act, reduce):
    """Local partial aggregate -> (all-reduce) -> bias + self loop + activation; mirrors _RelGraphConvBdd."""
    msg = orgcn._messages(x, src, et, norm, {'weight': w}, 'bdd', NB)
    agg = torch.zeros(x.shape[0], msg.shape[1]).index_add(0, dst, msg)
    agg = reduce(agg)
    h = agg + b + x @ lw
    return act(h) if act is not None else h


def loss_on(params, src, dst, et, norm, trip, labels, reduce):
    x = params['emb']
    h1 = layer(x, src, dst, et, norm, params['w1'], params['b1'], params['l1'], torch.relu, reduce)
    h2 = layer(h1, src, dst, et, norm, params['w2'], params['b2'], params['l2'], None, reduce)
    score = okg.distmult_score(h2, params['w_rel'], trip)
    return torch.nn.functional.binary_cross_entropy_with_logits(score, labels)


def reference_grads():
    src, dst, et, trip, labels, params = make_problem()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    src_t, dst_t, et_t = (torch.from_numpy(a) for a in (src, dst, et))
    deg = torch.bincount(dst_t, minlength=N).float()
    norm = torch.where(deg > 0, 1.0 / deg.clamp(min=1), torch.zeros_like(deg))[dst_t].view(-1, 1)
    half = T // 2
    total = 0
    for r in range(2):
        tr = torch.from_numpy(trip[r * half:(r + 1) * half])
        lb = torch.from_numpy(labels[r * half:(r + 1) * half])
        total = total + loss_on(p, src_t, dst_t, et_t, norm, tr, lb, lambda t: t) / 2
    total.backward()
    return {k: v.grad.clone() for k, v in p.items()}, float(total.detach())


def worker(rank, world, port, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gcn_vae_amd import distributed as gdist
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        src, dst, et, trip, labels, params = make_problem()
        mine, (lo, hi) = gdist.shard_edges_by_relation(et, R, world, rank)
        counts = [None] * world
        dist.all_gather_object(counts, (len(mine), lo, hi))
        assert sum(c[0] for c in counts) == E and counts[0][2] == counts[1][1]          # disjoint and complete
        assert abs(counts[0][0] - counts[1][0]) < 0.35 * E                                # roughly balanced
        src_l, dst_l, et_l = (torch.from_numpy(a[mine]) for a in (src, dst, et))
        node_norm = gdist.global_in_degree_norm(dst_l, N)
        deg = torch.bincount(torch.from_numpy(dst), minlength=N).float()
        assert torch.allclose(node_norm, torch.where(deg > 0, 1.0 / deg.clamp(min=1), torch.zeros_like(deg)))
        norm = node_norm[dst_l].view(-1, 1)
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        half = T // 2
        tr = torch.from_numpy(trip[rank * half:(rank + 1) * half])
        lb = torch.from_numpy(labels[rank * half:(rank + 1) * half])
        loss = loss_on(p, src_l, dst_l, et_l, norm, tr, lb, lambda t: gdist.AllReduceSum.apply(t, None))
        loss.backward()
        plist = list(p.values())
        gdist.average_gradients(plist)
        flat = torch.cat([v.grad.reshape(-1) for v in plist]).clone()
        # average_flat on a raw arena gives the same as the per-tensor path
        arena = torch.full((7,), float(rank + 1))
        gdist.average_flat(arena)
        assert torch.allclose(arena, torch.full((7,), 1.5))
        hook = gdist.make_reduce_hook()
        t = torch.full((3,), float(rank))
        for h in (gdist.make_reduce_hook(), gdist.make_reduce_hook(async_op=False)):
            t = torch.full((3,), float(rank))
            h(t).wait()
            assert torch.equal(t, torch.full((3,), 1.0))                           # in place, summed over ranks
        mean_loss = torch.tensor([float(loss)])
        dist.all_reduce(mean_loss)
        out_q.put((rank, {k: v.grad.clone().numpy() for k, v in p.items()}, float(mean_loss) / world, flat.numpy()))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_edge_sharded_training_step_equals_single_process():
    ref, ref_loss = reference_grads()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, grads, mean_loss, _ in results:
        assert abs(mean_loss - ref_loss) < 1e-5
        for k, g in grads.items():
            np.testing.assert_allclose(g, ref[k].numpy(), rtol=1e-4, atol=1e-6, err_msg=f'rank {rank} {k}')
    np.testing.assert_array_equal(results[0][3], results[1][3])       # replicas stay bit-identical


def test_shard_edges_by_relation_partitions():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gcn_vae_amd import distributed as gdist
    rs = np.random.RandomState(1)
    et = rs.randint(0, 474, 50000)
    for world in (1, 2, 4, 8):
        seen = np.zeros(len(et), dtype=int)
        sizes = []
        for r in range(world):
            ids, (lo, hi) = gdist.shard_edges_by_relation(et, 474, world, r)
            seen[ids] += 1
            sizes.append(len(ids))
            assert ((et[ids] >= lo) & (et[ids] < hi)).all()
        assert (seen == 1).all()
        assert max(sizes) - min(sizes) < 0.1 * len(et)
    # degenerate: fewer relations than ranks -> empty shards allowed, still a partition
    et2 = np.zeros(100, dtype=int)
    tot = sum(len(gdist.shard_edges_by_relation(et2, 1, 4, r)[0]) for r in range(4))
    assert tot == 100
