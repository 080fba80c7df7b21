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


def collective_timeout():
    """Seconds a collective may take before the process group gives up on it (GV_DIST_TIMEOUT, default 300 -- torch's own default
    is 10 / 30 minutes): a rank that died or never arrived then costs every other rank that long at most, after which RCCL's
    watchdog (or gloo's wait) raises and the rank exits non-zero naming the phase it was in (bench.py, train.py)."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get('GV_DIST_TIMEOUT', '300')))


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
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank), timeout=collective_timeout())
        else:
            dist.init_process_group(backend, timeout=collective_timeout())
    return rank, local_rank, world


class _Done:
    def wait(self):
        return None


# A segments.SegmentedGraph that is recording the step (None otherwise): collectives are then issued BETWEEN hipGraph
# segments and recorded for replay.  Every collective on the step goes through start_collective.
RECORDER = None


def start_collective(start_fn):
    """``start_fn()`` issues a collective and returns an object with wait() (or None when it blocks).  Returns a handle
    with wait().  The tensors ``start_fn`` closes over must be the step's own (static under segmented capture)."""
    # work a backward pass left on the 'bwd' side stream (a MADE's weight-gradient products) is joined here: the collective may
    # read what it writes (the gradient arena), and under segmented capture a fork must be joined inside its own graph segment
    from . import ops as _ops
    _ops.backward_side_finish()
    if RECORDER is not None:
        return RECORDER.collective(start_fn)
    work = start_fn()
    return work if work is not None else _Done()


def make_reduce_hook(group=None, async_op=True):
    """Callable for ``RelGraphConv.reduce_hook``: starts an in-place sum of ``t`` over the edge shards and
    returns a handle whose ``wait()`` orders the current stream behind it.  With ``async_op`` the collective
    runs on RCCL's own stream, so whatever the caller enqueues between the call and ``wait()`` (the layer's
    self-loop GEMM in forward; bias / loop-weight gradients and the loop GEMM in backward) overlaps with it."""
    def hook(t):
        return start_collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=bool(async_op)))
    return hook


class AllReduceSum(torch.autograd.Function):
    """y = sum_p x_p on every rank; backward: grad_x_p = sum_q grad_y_q."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        if dist.is_initialized():
            start_collective(lambda: dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group, async_op=True)).wait()
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        if dist.is_initialized():
            start_collective(lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group, async_op=True)).wait()
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
    if dist.is_initialized():
        start_collective(lambda: dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group, async_op=True)).wait()
        if world > 1:
            flat_grads.div_(world)


def arena_order(model):
    """The trainable parameters of a LinkPredict(KGVAE) model ordered by when backward FINISHES them, last-finished
    first: entity table, layer 1, layer 2, then everything the loss head and the flows own.  A FlatAdam built on this
    order lets BucketedArenaReduce hand the tail of the arena to RCCL while layer 1's backward still runs.  (Pointer
    re-homing only: names, shapes and state_dict keys are untouched.)"""
    first = ('encoder.input_layer.', 'encoder.rconv_layer_1.', 'encoder.rconv_layer_2.')
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]

    def rank_of(k):
        for i, pre in enumerate(first):
            if k.startswith(pre):
                return i
        return len(first)
    return [p for _, p in sorted(named, key=lambda kp: rank_of(kp[0]))]       # stable: registration order within a group


class BucketedArenaReduce:
    """The parameter-gradient all-reduce of a flat gradient arena, cut where backward lets it start early.

    Backward finishes the parameters in the reverse of their registration order: the loss head / flows / layer 2 first,
    layer 1 and the entity table last.  A *milestone* is an activation whose gradient marks such a point: once autograd
    has produced the gradient of layer 2's INPUT, every parameter registered from layer 2 onwards is final, so that
    suffix of the arena is all-reduced asynchronously (RCCL's own stream) while layer 1's backward -- weight gradient,
    loop GEMMs and the K1^T aggregate, ~0.3 ms of kernels on BASELINE configs[1] -- still runs.  ``finish()`` (called where
    the one-shot ``average_flat`` / ``sum_flat`` was) reduces what is left -- the prefix that holds layer 1 and the
    entity table, final only when backward ends -- waits for everything and applies the 1/world scale once.

    ``offsets``: {parameter: first float of its slice}; FlatAdam exposes it as ``offsets``.  Works on any flat tensor
    (the world_size-2 gloo test drives it on CPU tensors).  Collectives go through ``start_collective`` so that a
    recording SegmentedGraph cuts its hipGraph segments around them.
    """

    def __init__(self, flat_grads, offsets, group=None, average=True):
        self.flat, self.offsets, self.group, self.average = flat_grads, dict(offsets), group, average
        self.reset()

    def reset(self):
        self.hi = self.flat.numel()          # floats [hi, end) are already handed to a collective
        self.handles = []
        self.log = []                        # (lo, hi) of every launched range, in launch order (tests read it)

    def _launch(self, lo):
        if lo >= self.hi or not dist.is_initialized():
            self.hi = min(self.hi, lo)
            return
        piece = self.flat[lo:self.hi]
        self.log.append((lo, self.hi))
        self.hi = lo
        self.handles.append(start_collective(
            lambda: dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def milestone(self, activation, first_param):
        """When the gradient of ``activation`` exists, everything from ``first_param``'s slice to the end of the not yet
        reduced arena is final: start its all-reduce.  Call during forward, on the tensor that enters the module whose
        first registered parameter is ``first_param``."""
        if not activation.requires_grad:
            return activation
        lo = self.offsets[first_param]

        def fire(grad):
            self._launch(lo)
            return grad
        activation.register_hook(fire)
        return activation

    def finish(self):
        self._launch(0)
        for h in self.handles:
            h.wait()
        if self.average and dist.is_initialized():
            world = dist.get_world_size(self.group)
            if world > 1:
                self.flat.div_(world)
        self.reset()


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


# ------------------------------------------------------------------------------------------------
# Destination-row partition (the "cheaper alternative to measure against"):
#
#   rank p OWNS a block of node rows and every edge that ends in them, so each R-GCN layer produces FINAL rows on
#   their owner -- no sum of partial aggregates -- and everything that is row-wise over nodes (self-loop GEMM,
#   epilogue, reparameterisation, KL) runs on 1/P of the rows instead of being replicated on every rank.
#
#   forward   layer 1 reads the replicated embedding lookup; its rows are ALL-GATHERED for layer 2 (under layer 2's
#             self-loop GEMM), z is all-gathered for the decoder (the rank's triplets touch arbitrary entities)
#   backward  dL/dz and dL/dh1 are partial over all rows on every rank -> REDUCE-SCATTER to the row owners (under the
#             loop-weight products and grad-W); the embedding / weight gradients are partial sums -> ONE all-reduce(sum)
#   bytes     2 all-gathers + 2 reduce-scatters of (N, h) + the flat parameter all-reduce  =  ~38 MB of all-reduce
#             equivalents per step at FB15k-237 / h = 200, against ~85 MB for the edge-block scheme above.
#
# Node rows are dealt to the ranks by in-degree (heaviest first to the lightest rank), so that every rank owns the same number of rows (within world-1) and
# about the same number of edges; positions are  rank * slot_rows + i,  a rank's real rows first, then a few all-zero
# padding rows so that all slots have the same size (all_gather_into_tensor / reduce_scatter_tensor).
class RowPartition:
    def __init__(self, world, rank, counts, group=None, native=None):
        self.world, self.rank = int(world), int(rank)
        self.counts = [int(c) for c in counts]
        if len(self.counts) != self.world:
            raise ValueError('one row count per rank')
        self.slot_rows = max(max(self.counts), 1)
        self.own_rows = self.counts[self.rank]
        self.row0 = self.rank * self.slot_rows
        self.total_rows = self.world * self.slot_rows
        self.real_rows = sum(self.counts)
        self.group = group
        if native is None:      # all_gather_into_tensor / reduce_scatter_tensor: RCCL yes, gloo no
            native = dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'nccl'
        self.native = bool(native)
        # PIPELINED exchanges (GV_DIST_ROW_CHUNKS = C > 1): a layer's output rows leave in C blocks of slot rows -- block k is
        # gathered while block k + 1 is still being aggregated -- and the backward's partial gradient of the table is
        # reduce-scattered the same way, block by block under the next block's K1^T (gather_rows / reduce_scatter_rows)
        self.chunks = max(1, int(os.environ.get('GV_DIST_ROW_CHUNKS', '1')))

    def chunk_bounds(self, chunks=None):
        """[(a, b)] slot-row ranges of the pipelined exchanges: the same on every rank (collectives need equal sizes)."""
        c = max(1, min(int(chunks or self.chunks), self.slot_rows))
        cuts = sorted(set([0] + [(self.slot_rows * k) // c for k in range(1, c)] + [self.slot_rows]))
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]

    def _views(self, full, a, b):
        return [full[r * self.slot_rows + a:r * self.slot_rows + b] for r in range(self.world)]

    def gather_rows(self, out_full, x_slot, a, b):
        """Rows [a, b) of EVERY rank's slot into their places of the full table (out_full[r * slot + a : r * slot + b] <- rank r's
        x_slot[a:b]).  Returns a handle with wait()."""
        if self.world == 1:
            out_full[a:b].copy_(x_slot[a:b])
            return _Done()
        mine = x_slot[a:b].contiguous()
        if self.native:      # RCCL: the list form (the outputs are row ranges of one buffer, a stride apart)
            views = self._views(out_full, a, b)
            return start_collective(lambda: dist.all_gather(views, mine, group=self.group, async_op=True))
        tmp = torch.zeros(self.world, b - a, x_slot.shape[1], dtype=x_slot.dtype, device=x_slot.device)      # functional fallback (gloo)
        tmp[self.rank].copy_(mine)
        start_collective(lambda: dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
        for r, v in enumerate(self._views(out_full, a, b)):
            v.copy_(tmp[r])
        return _Done()

    def reduce_scatter_rows(self, out_slot, g_full, a, b):
        """out_slot[a:b] <- sum over ranks of THIS rank's rows [a, b) of g_full (g_full[rank * slot + a : rank * slot + b])."""
        if self.world == 1:
            out_slot[a:b].copy_(g_full[a:b])
            return _Done()
        if self.native:
            ins = [v.contiguous() for v in self._views(g_full, a, b)]
            out = out_slot[a:b]
            return start_collective(lambda: dist.reduce_scatter(out, ins, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        tmp = torch.stack([v for v in self._views(g_full, a, b)], 0).contiguous()                              # functional fallback (gloo)
        start_collective(lambda: dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
        out_slot[a:b].copy_(tmp[self.rank])
        return _Done()

    def all_gather(self, out_full, x_slot):
        """out_full (world*slot, h) <- every rank's slot (slot, h).  Returns a handle with wait()."""
        if tuple(out_full.shape) != (self.total_rows, x_slot.shape[1]) or x_slot.shape[0] != self.slot_rows:
            raise ValueError(f'all_gather: shapes {tuple(out_full.shape)} / {tuple(x_slot.shape)} do not fit '
                             f'{self.world} slots of {self.slot_rows} rows')
        if self.native:
            x_slot = x_slot.contiguous()
            return start_collective(lambda: dist.all_gather_into_tensor(out_full, x_slot, group=self.group, async_op=True))
        if self.world == 1:
            out_full.copy_(x_slot)
            return _Done()
        out_full.zero_()                                     # functional fallback (gloo): a sum of disjoint slots
        out_full[self.row0:self.row0 + self.slot_rows].copy_(x_slot)
        return start_collective(lambda: dist.all_reduce(out_full, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_scatter(self, out_slot, g_full):
        """out_slot (slot, h) <- this rank's slot of the sum over ranks of g_full (world*slot, h)."""
        if g_full.shape[0] != self.total_rows or tuple(out_slot.shape) != (self.slot_rows, g_full.shape[1]):
            raise ValueError('reduce_scatter: shape mismatch')
        if self.native:
            g_full = g_full.contiguous()
            return start_collective(lambda: dist.reduce_scatter_tensor(out_slot, g_full, op=dist.ReduceOp.SUM,
                                                                       group=self.group, async_op=True))
        if self.world == 1:
            out_slot.copy_(g_full)
            return _Done()
        tmp = g_full.clone()                                 # functional fallback (gloo)
        start_collective(lambda: dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
        out_slot.copy_(tmp[self.row0:self.row0 + self.slot_rows])
        return _Done()

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class AllGatherRows(torch.autograd.Function):
    """full (world*slot, h) = all-gather of the ranks' slots; backward = reduce-scatter(sum) of the gradient."""

    @staticmethod
    def forward(ctx, x_slot, part):
        ctx.part = part
        out = torch.empty(part.total_rows, x_slot.shape[1], dtype=x_slot.dtype, device=x_slot.device)
        part.all_gather(out, x_slot).wait()
        return out

    @staticmethod
    def backward(ctx, g):
        part = ctx.part
        own = torch.empty(part.slot_rows, g.shape[1], dtype=g.dtype, device=g.device)
        part.reduce_scatter(own, g.contiguous()).wait()
        return own, None


def plan_row_partition(in_degree, world, greedy_nodes=32768):
    """Deal the nodes to ``world`` ranks: equal row counts (within a row or two) and near-equal edge counts.  The
    heaviest ``greedy_nodes`` nodes go one by one, in order of decreasing in-degree, to the rank with the fewest edges
    so far that still has room (longest-processing-time rule: a hub of a Zipf-like degree law is worth thousands of
    tail nodes); the light tail is dealt in snake order.  Returns (pos_of_node int64 [N], node_of_pos int64
    [world*slot] with padding positions -1, counts [world])."""
    import heapq
    deg = np.asarray(in_degree).astype(np.int64).reshape(-1)
    n = deg.shape[0]
    order = np.argsort(-deg, kind='stable')
    owner = np.empty(n, dtype=np.int64)
    k_greedy = min(n, int(greedy_nodes))
    cap = -(-k_greedy // world)
    load, count = [0] * world, [0] * world
    heap = [(0, p) for p in range(world)]
    for node in order[:k_greedy]:
        while True:
            l, p = heapq.heappop(heap)
            if count[p] < cap:
                break
        owner[node] = p
        load[p] = l + int(deg[node])
        count[p] += 1
        if count[p] < cap:
            heapq.heappush(heap, (load[p], p))
    rest = order[k_greedy:]
    if rest.size:
        by_load = np.argsort(np.asarray(load), kind='stable')              # lightest rank first
        i = np.arange(rest.size)
        k, rnd = i % world, i // world
        owner[rest] = by_load[np.where(rnd % 2 == 0, k, world - 1 - k)]
    counts = np.bincount(owner, minlength=world) if n else np.zeros(world, dtype=np.int64)
    slot = max(int(counts.max()) if n else 0, 1)
    pos_of_node = np.empty(n, dtype=np.int64)
    node_of_pos = np.full(world * slot, -1, dtype=np.int64)
    for p in range(world):
        mine = np.nonzero(owner == p)[0]                     # ascending node id inside a rank's block
        pos_of_node[mine] = p * slot + np.arange(mine.shape[0])
        node_of_pos[p * slot:p * slot + mine.shape[0]] = mine
    return pos_of_node, node_of_pos, [int(c) for c in counts]


class RowBlockGraph:
    """Graph handle of ONE rank's row block for the encoder: destinations are local rows, sources are positions in the
    full table.  Edges are kept in the reference's (dst, src, rel) order (kgvae/utils.py:146-147)."""

    def __init__(self, part, src_pos, dst_local, device):
        from . import ops
        self.part = part
        self._n = part.own_rows
        self.num_edges = int(src_pos.numel())
        self._index = ops.GraphIndex(src_pos.to(device), dst_local.to(device), part.own_rows, dst_sorted=True,
                                     num_src_nodes=part.total_rows)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return self.num_edges

    def device_index(self, device):
        return self._index


def build_row_block(part, pos_of_node, src, dst, rel, node_norm, device):
    """This rank's block of the union graph: the edges (node ids ``src -> dst``, relation ``rel``, device or host
    int64) whose destination the rank owns, relabelled to (position, local row) and sorted by (dst, src, rel).
    Returns (RowBlockGraph, etypes int64 [E_loc], edge_norm fp32 [E_loc, 1])."""
    dev = torch.device(device)
    pos = torch.as_tensor(pos_of_node, dtype=torch.int64).to(dev)
    src, dst, rel = (torch.as_tensor(t, dtype=torch.int64).to(dev) for t in (src, dst, rel))
    pd = pos[dst]
    mine = (pd >= part.row0) & (pd < part.row0 + part.slot_rows)
    ps, dl, rl, dn = pos[src[mine]], pd[mine] - part.row0, rel[mine], dst[mine]
    n_rel = int(rel.max()) + 1 if rel.numel() else 1
    key = (dl * part.total_rows + ps) * n_rel + rl
    order = torch.argsort(key, stable=True)
    ps, dl, rl, dn = ps[order], dl[order], rl[order], dn[order]
    norm = torch.as_tensor(node_norm, dtype=torch.float32).to(dev)
    g = RowBlockGraph(part, ps, dl, dev)
    return g, rl.contiguous(), norm[dn].view(-1, 1).contiguous()


def make_row_partition(in_degree, world, rank, group=None, native=None):
    """plan_row_partition + RowPartition; the plan is attached as ``pos_of_node`` / ``node_of_pos`` (numpy int64,
    padding positions -1) and ``real_positions`` (the positions that hold a node)."""
    pos_of_node, node_of_pos, counts = plan_row_partition(in_degree, world)
    part = RowPartition(world, rank, counts, group=group, native=native)
    part.pos_of_node, part.node_of_pos = pos_of_node, node_of_pos
    part.real_positions = np.nonzero(node_of_pos >= 0)[0]
    return part


def sum_flat(flat_grads, group=None):
    """Row partition: every parameter gradient is a partial sum over the ranks' rows / triplet shares -> one
    all-reduce(sum) of the FlatAdam gradient arena, no division."""
    if dist.is_initialized():
        start_collective(lambda: dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group, async_op=True)).wait()


# ---- the optimiser sharded over the ranks (round 4; SURVEY 8(e): "replicated params' grads all-reduced once per step" is what this
# replaces) -----------------------------------------------------------------------------------------------------------------
# With replicated parameters every rank all-reduces the whole gradient arena and runs clip + Adam on all of it.  Sharded: the
# arena is cut into `world` equal pieces;   reduce-scatter(sum) of the gradients  ->  the global gradient norm from ONE all-reduce
# of the pieces' sums of squares  ->  clip + Adam on the rank's piece (moments exist for that piece only)  ->  all-gather of the
# updated parameters.  Same bytes on the links as the all-reduce (which is a reduce-scatter + all-gather), but the second half
# now carries PARAMETERS, so the update itself is done once instead of `world` times and the moments take 1/world of the memory.
class ShardedArenaStep:
    """The communication + update pattern over flat tensors of any device.  ``sumsq(g) -> 0-d tensor`` and
    ``adam(p, g, m, v, sumsq, step_t)`` are the local kernels (HIP: gv_mean_sq / gv_adam_step; the CPU tests pass torch
    restatements)."""

    def __init__(self, flat_p, flat_g, sumsq, adam, group=None, average=True, native=None):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if flat_p.numel() != flat_g.numel() or flat_p.numel() % self.world:
            raise ValueError(f'ShardedArenaStep: arenas of {flat_p.numel()} / {flat_g.numel()} entries, not a multiple of {self.world} ranks')
        self.flat_p, self.flat_g = flat_p, flat_g
        self.n = flat_p.numel() // self.world
        self.lo = self.rank * self.n
        self.m = torch.zeros(self.n, dtype=flat_p.dtype, device=flat_p.device)
        self.v = torch.zeros(self.n, dtype=flat_p.dtype, device=flat_p.device)
        self.g_piece = torch.zeros(self.n, dtype=flat_p.dtype, device=flat_p.device)
        self.step_t = torch.zeros((), dtype=torch.float32, device=flat_p.device)
        self.total_sumsq = torch.zeros((), dtype=torch.float32, device=flat_p.device)
        self._sumsq, self._adam, self.average = sumsq, adam, bool(average)
        self._gather = None
        if native is None:
            native = self.world > 1 and dist.get_backend(group) == 'nccl'
        self.native = bool(native)

    def step(self):
        p_piece = self.flat_p[self.lo:self.lo + self.n]
        if self.world == 1:
            self.g_piece.copy_(self.flat_g)
        elif self.native:
            start_collective(lambda: dist.reduce_scatter_tensor(self.g_piece, self.flat_g, op=dist.ReduceOp.SUM, group=self.group,
                                                                async_op=True)).wait()
        else:           # functional fallback (gloo has no reduce-scatter): all-reduce, keep the rank's piece
            start_collective(lambda: dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
            self.g_piece.copy_(self.flat_g[self.lo:self.lo + self.n])
        if self.average and self.world > 1:
            self.g_piece.mul_(1.0 / self.world)
        self.total_sumsq.copy_(self._sumsq(self.g_piece))
        if self.world > 1:          # the norm clip_grad_norm_ takes over ALL parameters: the pieces' sums of squares, summed
            start_collective(lambda: dist.all_reduce(self.total_sumsq, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
        self.step_t += 1
        self._adam(p_piece, self.g_piece, self.m, self.v, self.total_sumsq, self.step_t)
        if self.world > 1:
            if self.native:
                self.g_piece.copy_(p_piece)          # (the gathered buffer must not overlap the piece it is gathered from; g_piece is free now)
                start_collective(lambda: dist.all_gather_into_tensor(self.flat_p, self.g_piece, group=self.group, async_op=True)).wait()
            else:       # functional fallback (gloo): a sum of disjoint pieces -- in a buffer of its own, never in the parameter arena
                # itself: while a SegmentedGraph records the step the kernels around an (eagerly issued) collective are captured, not
                # run, and a collective that sums the UN-zeroed arenas in place would double the parameters once, at capture time
                if self._gather is None:
                    self._gather = torch.zeros_like(self.flat_p)
                self._gather.zero_()
                self._gather[self.lo:self.lo + self.n].copy_(p_piece)
                start_collective(lambda: dist.all_reduce(self._gather, op=dist.ReduceOp.SUM, group=self.group, async_op=True)).wait()
                self.flat_p.copy_(self._gather)
        self.flat_g.zero_()


def ShardedFlatAdam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, group=None, average=True):
    """optim.FlatAdam whose step is sharded over the ranks (ShardedArenaStep on the HIP kernels): same arenas for parameters and
    gradients (backward kernels still add straight into the gradient arena), moments for the rank's piece only.  Returns the
    FlatAdam with ``step`` rebound; ``snapshot`` / ``restore`` are FlatAdam's own and work on the rank's piece through the aliased
    ``exp_avg`` / ``exp_avg_sq`` / ``step_t`` attributes.  ``max_grad_norm`` is required (the clip is part of the update)."""
    from . import lib
    from .lib import ptr
    from .optim import FlatAdam
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    opt = FlatAdam(params, lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm, total_multiple=world * FlatAdam.ALIGN, moments=False)
    ws = torch.empty(1024, dtype=torch.float32, device=opt.flat_p.device)

    def sumsq(g):
        out = torch.empty((), dtype=torch.float32, device=g.device)
        lib.call('gv_mean_sq', ptr(g), g.numel(), 1.0, ptr(out), ptr(ws), 0, lib.stream())
        return out

    def adam(p, g, m, v, total, step_t):
        lib.call('gv_adam_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(total), float(max_grad_norm or 0.0), float(lr),
                 float(betas[0]), float(betas[1]), float(eps), ptr(step_t), lib.stream())

    sh = ShardedArenaStep(opt.flat_p, opt.flat_g, sumsq, adam, group=group, average=average)
    opt.sharded = sh
    opt.exp_avg, opt.exp_avg_sq, opt.step_t, opt.sumsq = sh.m, sh.v, sh.step_t, sh.total_sumsq

    def step():
        from . import ops
        ops.backward_side_finish()
        sh.step()
        opt._clean = True
    opt.step = step
    return opt
