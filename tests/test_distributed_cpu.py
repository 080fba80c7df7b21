"""world_size-2 gloo tests (CPU) of the multi-GPU scheme in gcn-vae_amd/distributed.py.

The exchange pattern (edge-block sharding by relation, all-reduce(sum) of partial node aggregates in
forward, all-reduce(sum) of their gradient in backward, one averaged all-reduce of parameter gradients)
is device independent; here the local arithmetic is the CPU oracle and the result must equal a
single-process run on the union graph with loss = mean of the ranks' losses."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import kgvae as okg
from oracle import rgcn as orgcn

N, R, H, NB, E, T = 60, 8, 8, 4, 500, 120
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def loss_on(params, src, dst, et, norm, trip, labels, reduce, tap=None):
    x = params['emb']
    h1 = layer(x, src, dst, et, norm, params['w1'], params['b1'], params['l1'], torch.relu, reduce)
    if tap is not None:
        h1 = tap(h1)
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
        # the same step with the gradient arena reduced in two pieces: layer 2 + decoder as soon as dL/dh1 exists (under
        # layer 1's backward), the rest at the end -- equal to the one-shot average, bit for bit
        sizes = [v.numel() for v in params.values()]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        flat_g = torch.zeros(int(offs[-1]))
        p2 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        for (k, v), o in zip(p2.items(), offs):
            v.grad = flat_g[o:o + v.numel()].view_as(v)
        red = gdist.BucketedArenaReduce(flat_g, {v: int(o) for v, o in zip(p2.values(), offs)})
        loss2 = loss_on(p2, src_l, dst_l, et_l, norm, tr, lb, lambda t: gdist.AllReduceSum.apply(t, None),
                        tap=lambda h1: red.milestone(h1, p2['w2']))
        loss2.backward()
        off_w2 = int(offs[list(p2).index('w2')])
        assert red.log == [(off_w2, int(offs[-1]))], red.log              # the suffix went out DURING backward
        red.finish()
        assert torch.equal(flat_g, flat), float((flat_g - flat).abs().max())
        assert all(v.grad.data_ptr() == flat_g[o:].data_ptr() for v, o in zip(p2.values(), offs))      # grads still alias the arena
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


# ------------------------------------------------------------------------------------------------
# destination-row partition (distributed.RowPartition): the same device-independent checks
def test_plan_row_partition_balances_rows_and_edges():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gcn_vae_amd import distributed as gdist
    rs = np.random.RandomState(0)
    n = 14541
    p = (np.arange(n) + 1.0) ** -0.8
    deg = np.bincount(rs.choice(n, size=544230, p=p / p.sum()), minlength=n)           # FB15k-237-like hub skew
    for world in (1, 2, 3, 8):
        pos_of_node, node_of_pos, counts = gdist.plan_row_partition(deg, world)
        slot = max(counts)
        assert sum(counts) == n and max(counts) - min(counts) < max(world, 2) and len(node_of_pos) == world * slot
        real = node_of_pos >= 0
        assert real.sum() == n and (np.sort(node_of_pos[real]) == np.arange(n)).all()      # a bijection onto positions
        assert (node_of_pos[pos_of_node] == np.arange(n)).all()
        owner = pos_of_node // slot
        edges = np.bincount(owner, weights=deg, minlength=world)
        assert edges.max() <= 1.02 * edges.mean(), (world, edges)                        # near-equal edge counts despite the hubs
        for r in range(world):            # a rank's real rows come first in its slot, ascending node id
            blk = node_of_pos[r * slot:(r + 1) * slot]
            assert (blk[:counts[r]] >= 0).all() and (blk[counts[r]:] == -1).all() and (np.diff(blk[:counts[r]]) > 0).all()
    part = gdist.RowPartition(3, 1, [5, 4, 4], native=False)
    assert (part.slot_rows, part.own_rows, part.row0, part.total_rows, part.real_rows) == (5, 4, 5, 15, 13)


def rows_reference():
    """Single process, whole graph: L = mean over the two triplet halves of BCE + 0.1 mean(h2^2)."""
    src, dst, et, trip, labels, params = make_problem()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    src_t, dst_t, et_t = (torch.from_numpy(a) for a in (src, dst, et))
    deg = torch.bincount(dst_t, minlength=N).float()
    norm = torch.where(deg > 0, 1.0 / deg.clamp(min=1), torch.zeros_like(deg))[dst_t].view(-1, 1)
    x = p['emb']
    h1 = layer(x, src_t, dst_t, et_t, norm, p['w1'], p['b1'], p['l1'], torch.relu, lambda t: t)
    h2 = layer(h1, src_t, dst_t, et_t, norm, p['w2'], p['b2'], p['l2'], None, lambda t: t)
    score = okg.distmult_score(h2, p['w_rel'], torch.from_numpy(trip))
    loss = torch.nn.functional.binary_cross_entropy_with_logits(score, torch.from_numpy(labels)) + 0.1 * h2.pow(2).mean()
    loss.backward()
    return {k: v.grad.clone() for k, v in p.items()}, float(loss.detach())


def rows_worker(rank, world, port, out_q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gcn_vae_amd import distributed as gdist
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        src, dst, et, trip, labels, params = make_problem()
        deg = np.bincount(dst, minlength=N)
        part = gdist.make_row_partition(deg, world, rank)
        assert not part.native and part.real_rows == N
        c, slot, row0 = part.own_rows, part.slot_rows, part.row0
        pos = torch.from_numpy(part.pos_of_node)
        nop = torch.from_numpy(np.maximum(part.node_of_pos, 0))
        # the collectives themselves (gloo fallbacks)
        mine = torch.full((slot, 2), float(rank + 1))
        full = torch.empty(world * slot, 2)
        part.all_gather(full, mine).wait()
        assert all(torch.equal(full[r * slot:(r + 1) * slot], torch.full((slot, 2), float(r + 1))) for r in range(world))
        own = torch.empty(slot, 2)
        part.reduce_scatter(own, full * (rank + 1)).wait()
        assert torch.equal(own, torch.full((slot, 2), float((rank + 1) * 3)))
        # the pipelined exchanges' block forms: rows [a, b) of every slot gathered into their places / reduce-scattered from them
        bounds = part.chunk_bounds(3)
        assert bounds[0][0] == 0 and bounds[-1][1] == slot and all(bounds[i][1] == bounds[i + 1][0] for i in range(len(bounds) - 1))
        mine = torch.arange(slot * 2, dtype=torch.float32).view(slot, 2) + 100.0 * rank
        full2 = torch.full((world * slot, 2), -1.0)
        for a, b in bounds:
            part.gather_rows(full2, mine, a, b).wait()
        for r in range(world):
            assert torch.equal(full2[r * slot:(r + 1) * slot], torch.arange(slot * 2, dtype=torch.float32).view(slot, 2) + 100.0 * r)
        own2 = torch.full((slot, 2), -1.0)
        gsrc = (torch.arange(world * slot * 2, dtype=torch.float32).view(world * slot, 2)) * (rank + 1)
        for a, b in bounds:
            part.reduce_scatter_rows(own2, gsrc, a, b).wait()
        want = torch.arange(world * slot * 2, dtype=torch.float32).view(world * slot, 2)[rank * slot:(rank + 1) * slot] * sum(range(1, world + 1))
        assert torch.equal(own2, want)
        # this rank's edges: destination in its row block; sources are positions in the full table
        ps, pd = pos[torch.from_numpy(src)], pos[torch.from_numpy(dst)]
        sel = (pd >= row0) & (pd < row0 + slot)
        ps, dl, et_l = ps[sel], pd[sel] - row0, torch.from_numpy(et)[sel]
        node_norm = torch.where(torch.from_numpy(deg) > 0, 1.0 / torch.from_numpy(deg).clamp(min=1).float(), torch.zeros(N))
        norm = node_norm[torch.from_numpy(dst)[sel]].view(-1, 1)
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}

        def layer_rows(x_full, w, b, lw, act):
            msg = orgcn._messages(x_full, ps, et_l, norm, {'weight': w}, 'bdd', NB)
            agg = torch.zeros(c, msg.shape[1]).index_add(0, dl, msg)
            h = agg + b + x_full[row0:row0 + c] @ lw
            return act(h) if act is not None else h

        def pad(t):
            return torch.cat([t, torch.zeros(slot - c, t.shape[1])]) if c < slot else t

        x0 = p['emb'][nop]                                                # replicated lookup of all positions
        h1 = gdist.AllGatherRows.apply(pad(layer_rows(x0, p['w1'], p['b1'], p['l1'], torch.relu)), part)
        h2_own = layer_rows(h1, p['w2'], p['b2'], p['l2'], None)
        h2 = gdist.AllGatherRows.apply(pad(h2_own), part)
        half = T // world
        tr = torch.from_numpy(trip[rank * half:(rank + 1) * half])
        tr_pos = torch.stack([pos[tr[:, 0]], tr[:, 1], pos[tr[:, 2]]], 1)
        lb = torch.from_numpy(labels[rank * half:(rank + 1) * half])
        pred = torch.nn.functional.binary_cross_entropy_with_logits(okg.distmult_score(h2, p['w_rel'], tr_pos), lb)
        # the rank's SHARE of the loss: replicated terms / world, row-wise terms over its own rows with the global mean's weight
        share = pred / world + 0.1 * h2_own.pow(2).sum() / (N * h2_own.shape[1])
        share.backward()
        arena = torch.cat([v.grad.reshape(-1) for v in p.values()])
        gdist.sum_flat(arena)
        total = share.detach().reshape(1).clone()
        dist.all_reduce(total)
        out_q.put((rank, arena.numpy(), float(total)))
    finally:
        dist.destroy_process_group()


def test_row_partitioned_training_step_equals_single_process():
    ref, ref_loss = rows_reference()
    ref_flat = torch.cat([ref[k].reshape(-1) for k in ref]).numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, arena, total in results:
        assert abs(total - ref_loss) < 1e-5
        np.testing.assert_allclose(arena, ref_flat, rtol=1e-4, atol=1e-6, err_msg=f'rank {rank}')
    np.testing.assert_array_equal(results[0][1], results[1][1])


# ---- the sharded optimiser (distributed.ShardedArenaStep): reduce-scatter -> norm -> clip + Adam on a piece -> all-gather -----
def _torch_sumsq(g):
    return (g.double() * g.double()).sum().float()


def _torch_adam(lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, max_norm=1.0):
    """torch restatement of gv_adam_step (csrc/k_elem.hip k_adam): clip coefficient from the TOTAL sum of squares, torch.optim.Adam's update."""
    def adam(p, g, m, v, total, step_t):
        clip = torch.clamp(max_norm / (total.sqrt() + 1e-6), max=1.0)
        gi = g * clip
        m.mul_(b1).add_(gi, alpha=1 - b1)
        v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
        t = float(step_t)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        p.sub_((lr / bc1) * m / (v.sqrt() / (bc2 ** 0.5) + eps))
    return adam


def sharded_worker(rank, world, port, out_q, steps):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import gcn_vae_amd  # noqa: F401
        from gcn_vae_amd import distributed as gdist
        n = 64 * world * 3
        gen = torch.Generator().manual_seed(0)
        p0 = torch.randn(n, generator=gen)
        flat_p, flat_g = p0.clone(), torch.zeros(n)
        sh = gdist.ShardedArenaStep(flat_p, flat_g, _torch_sumsq, _torch_adam(), average=True)
        assert sh.n == n // world and sh.m.numel() == n // world          # moments for the rank's piece only
        for s in range(steps):
            flat_g.copy_(torch.randn(n, generator=torch.Generator().manual_seed(100 * s + rank)) * (3.0 if s == 0 else 0.01))
            sh.step()
            assert float(flat_g.abs().max()) == 0.0                          # the step consumes the gradient arena
        out_q.put((rank, flat_p.numpy().copy(), float(sh.total_sumsq)))
    finally:
        dist.destroy_process_group()


def test_sharded_optimiser_equals_the_single_process_update():
    """Two gloo ranks, each with its own gradient arena: after ShardedArenaStep.step() every rank holds the SAME parameters, and
    they are the parameters one process gets from the averaged gradient with clip_grad_norm_ over the whole arena + Adam --
    bit for bit (the pieces' updates are element-wise; the norm is a sum of two piece sums in both runs), with the clip active
    (step 0: large gradients) and inactive (later steps)."""
    world, steps = 2, 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=sharded_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: the same arithmetic on the whole arena, the norm summed piece by piece as the ranks do
    n = 64 * world * 3
    p_ref = torch.randn(n, generator=torch.Generator().manual_seed(0))
    m, v, adam = torch.zeros(n), torch.zeros(n), _torch_adam()
    for s in range(steps):
        gs = [torch.randn(n, generator=torch.Generator().manual_seed(100 * s + r)) * (3.0 if s == 0 else 0.01) for r in range(world)]
        g = (gs[0] + gs[1]) * (1.0 / world)
        total = _torch_sumsq(g[:n // 2]) + _torch_sumsq(g[n // 2:])
        if s == 0:
            assert float(total.sqrt()) > 1.0          # the clip is active in the first step
        adam(p_ref, g, m, v, total, torch.tensor(float(s + 1)))
    np.testing.assert_array_equal(results[0][1], results[1][1])
    np.testing.assert_array_equal(results[0][1], p_ref.numpy())


def _bench_env():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'GV_DIST_BACKEND'):
        env.pop(k, None)
    return env


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own 2-rank torch.distributed.run child (before any HIP
    call), the ranks meet over 127.0.0.1 (gloo on this GPU-less box), rank 0's one JSON line is relayed: n_gpus = 2,
    ranks_seen = 2, strong scaling by default.  --launch-check stops before the compute, which needs an MI355X."""
    import json
    bench = os.path.join(ROOT, 'bench.py')
    out = subprocess.run([sys.executable, bench, '--gpus', '2', '--launch-check'], cwd=ROOT, env=_bench_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['self_launched'] and d['scaling'] == 'strong'
    # under a launcher (the driver's form) the same flags do not start a second job
    port = free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), bench, '--gpus', '2', '--launch-check', '--scaling', 'weak']
    out = subprocess.run(cmd, cwd=ROOT, env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and not d['self_launched'] and d['scaling'] == 'weak'


def test_bench_self_launch_returns_the_childs_exit_code():
    """No GPU here: the ranks of a self-launched run refuse to compute (there is no CPU path) and bench.py exits non-zero
    with their message on stderr instead of printing a result line."""
    if torch.cuda.is_available():
        pytest.skip('needs a GPU-less box')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         cwd=ROOT, env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert 'needs an MI355X' in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]
