#!/usr/bin/env python3
"""bench.py -- edges/sec of the R-GCN-VAE forward+backward hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W      (no launcher: bench.py starts the N ranks itself, as a CHILD
                                                        torch.distributed.run job, before it touches the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--scaling strong|weak]
    python bench.py --config mb [--n-flows 3]          (the reference's mini-batch regime, one GPU)

Workloads (BASELINE.json configs, all seeded synthetics -- no dataset on disk, no network):
  c2 (default, configs[1]) FB15k-237-shaped: 14 541 entities, 237 relations = 474 directed edge types, 272 115 triplets
     with Zipf(0.8) endpoints => 544 230 directed edges, 2-layer R-GCN-VAE encoder (bdd, num_bases=100, emb_dim=200, fp32,
     dropout 0.2) on the FULL graph + DistMult decoder on 220 000 triplets (20 000 positives x (1 + 10 negatives))
  c3 (configs[2]) WN18RR-shaped: 40 943 entities, 11 relations, 86 835 triplets, emb_dim=200, num_bases=20, 3 IAF blocks,
     dense products with bf16 operands
  c4 (configs[3]) the c2 graph at emb_dim=500
  c5 (configs[4]) 1 M entities, 1 000 relations, 25 M triplets => 50 M directed edges, emb_dim=200 (generated on the device)
  mb (configs[1]'s data, the regime of kgvae/README.md:4-7) a fresh 20 000-triplet sample per step: E = 20 000 directed edges,
     T = 220 000 scored triplets; device sampler + index builders + step replayed as ONE hipGraph
One step = forward + loss (BCE + 0.01 reg + 1e-5 KL + 1.0 MMD) + backward + grad-clip + Adam, i.e. the reference's
t0..t2 span (kgvae/link_predict.py:222-229) with device synchronisation.
--scaling strong (default): ONE graph (seed 0), whatever the rank count: its directed edges are cut by RELATION across the
ranks (distributed.shard_edges_by_relation, north_star / BASELINE configs[3..4]) or by destination row (--partition row),
its triplets dealt round-robin.  --scaling weak: every rank brings its own edge block (seed = rank) and triplet slice;
the graph trained is the union of the blocks.
Node embeddings are exchanged over RCCL once per layer per direction, parameter gradients once per step.

Rank 0 prints ONE JSON line: value = directed edges of the trained graph x steps / max-over-ranks seconds.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MALL_GATHER_GBS = 8600.0  # indexed rows served from the Infinity Cache (same guide, "Indexed rows: gather into LDS")
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide; AMD's 5 PFLOP/s headline includes 2:1 sparsity)
F32_MFMA_PEAK_TFLOPS = 157.0     # dense fp32 MFMA peak (same guide)
L2_GATHER_GBS = 17800.0   # ... from an XCD's L2 (16.8-18.8 TB/s)

# BASELINE.json configs[1..4]; c2 is the configuration the metric is quoted on
CONFIGS = {
    'c2': dict(idx=1, label='FB15k-237-shaped synthetic full graph (BASELINE configs[1])', nodes=14541, rels=237,
               train=272115, hidden=200, bases=100, flows=0, gemm='f32', device_gen=False),
    'c3': dict(idx=2, label='WN18RR-shaped synthetic full graph + 3 IAF blocks (BASELINE configs[2])', nodes=40943, rels=11,
               train=86835, hidden=200, bases=20, flows=3, gemm='bf16', device_gen=False),
    'c4': dict(idx=3, label='FB15k-237-shaped synthetic full graph at emb_dim=500 (BASELINE configs[3])', nodes=14541, rels=237,
               train=272115, hidden=500, bases=100, flows=0, gemm='f32', device_gen=False),
    'c5': dict(idx=4, label='synthetic KG 1M entities / 50M directed edges / 1k relations (BASELINE configs[4])',
               nodes=1_000_000, rels=1000, train=25_000_000, hidden=200, bases=100, flows=0, gemm='f32', device_gen=True),
    # the regime the reference actually trains in (kgvae/README.md:4-7, kgvae/link_predict.py:200-236): every step samples
    # 20 000 triplets of the FB15k-237-shaped set, keeps half as the message-passing graph (E = 20 000 directed edges, N ~ 10 k)
    # and scores all 20 000 x (1 + 10 negatives) = 220 000 triplets; sampler + indices + step are ONE hipGraph replay
    'mb': dict(idx=1, label='FB15k-237-shaped synthetic, MINI-BATCH regime of kgvae/README.md:4-7 (SURVEY 8(d) C2(i))', nodes=14541,
               rels=237, train=272115, hidden=200, bases=100, flows=0, gemm='f32', device_gen=False),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=30)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--config', choices=sorted(CONFIGS), default='c2', help='BASELINE.json workload (see the module docstring)')
    p.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                   help='world size > 1: strong = ONE graph cut across the ranks (BASELINE configs[3..4]), weak = one edge '
                        'block per rank (the union graph grows with the rank count)')
    p.add_argument('--repeats', type=int, default=3,
                   help='timed regions of --steps steps each (the contract\'s EXACTLY K steps inside the bracket): `value` is their median, '
                        'the first region\'s figure is reported beside it')
    p.add_argument('--launch-check', action='store_true',
                   help='only start the ranks, form the process group (gloo when there is no GPU) and report how many ranks '
                        'met; no compute (tests/test_distributed_cpu.py runs it on CPU)')
    p.add_argument('--hidden', type=int, default=None, help='override the config\'s emb_dim')
    p.add_argument('--n-bases', type=int, default=None)
    p.add_argument('--n-flows', type=int, default=None)
    p.add_argument('--gemm-precision', choices=['f32', 'bf16'], default=None,
                   help="bf16: dense products with bf16 operands / fp32 accumulation (configs[2]'s precision)")
    p.add_argument('--no-check', action='store_true', help='skip the parity leg (one HIP step against the CPU oracle)')
    p.add_argument('--positives', type=int, default=20000)
    p.add_argument('--negative-sample', type=int, default=10)
    p.add_argument('--dropout', type=float, default=0.2)
    p.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a hipGraph')
    p.add_argument('--graph-collectives', action='store_true',
                   help='world size > 1: capture the RCCL collectives inside ONE hipGraph (default there: a chain of '
                        'hipGraph segments with the collectives launched eagerly between them)')
    p.add_argument('--no-segments', action='store_true',
                   help='world size > 1: launch every kernel eagerly instead of replaying hipGraph segments')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--rewarm-seconds', type=float, default=0.5,
                   help='untimed steps for this long behind the CPU-baseline leg, in front of the --warmup steps (the GPU clocks drop while '
                        'the CPU works; 0: none)')
    p.add_argument('--cpu-seconds', type=float, default=90.0,
                   help='budget of the CPU-oracle baseline leg (the default covers 3 warm-up + 20 timed oracle steps of the default workload)')
    p.add_argument('--force-dist', action='store_true', help='run the RCCL code path even at world size 1 (testing)')
    p.add_argument('--partition', choices=['auto', 'edge', 'row'], default='auto',
                   help='world size > 1: "edge" = edge-block sharding + all-reduce of node embeddings (north_star), "row" = '
                        'destination-row partition + all-gather / reduce-scatter (SURVEY 8e alternative), "auto" = time '
                        '--probe-steps steps of each during warm-up and run the faster one')
    p.add_argument('--sharded-adam', action='store_true',
                   help='world size > 1: distributed.ShardedFlatAdam -- reduce-scatter of the gradient arena, clip + Adam on 1/N of '
                        'it per rank, all-gather of the updated parameters -- instead of the all-reduce + replicated update')
    p.add_argument('--probe-steps', type=int, default=10)
    p.add_argument('--profile-steps', type=int, default=3, help='instrumented eager steps for the roofline figure')
    args = p.parse_args()
    cfg = CONFIGS[args.config]
    for name, key in (('hidden', 'hidden'), ('n_bases', 'bases'), ('n_flows', 'flows'), ('gemm_precision', 'gemm')):
        if getattr(args, name) is None:
            setattr(args, name, cfg[key])
    return args


class _Data:
    def __init__(self, num_nodes, num_rels, train):
        self.num_nodes, self.num_rels, self.train = num_nodes, num_rels, train


def directed_graph(cfg, seed, dev):
    """One synthetic knowledge graph as the reference hands it to the encoder (kgvae/utils.py:135-150): reverse edges
    added (relation id + num_rels), edges sorted by (dst, src, rel).  Returns CPU int64 tensors (src, dst, rel) and the
    (n, 3) triplets; the 50 M-edge configuration is generated and sorted on the device."""
    from gcn_vae_amd import sampling
    from gcn_vae_amd.data import synthetic_kg
    n, nr, t = cfg['nodes'], cfg['rels'], cfg['train']
    if not cfg['device_gen']:
        data = synthetic_kg(n, nr, t, seed=seed)
        g, rel, _ = sampling.build_test_graph(n, nr, data.train)
        src, dst = g.edges()
        return src, dst, torch.from_numpy(rel).long(), data.train
    gen = torch.Generator(device=dev).manual_seed(seed)
    # Zipf(0.8) endpoints by inverse CDF (P(i) ~ (i+1)^-0.8  =>  i ~ n u^5), uniform relations
    s = (torch.rand(t, device=dev, generator=gen) ** 5 * n).long().clamp_(max=n - 1)
    o = (torch.rand(t, device=dev, generator=gen) ** 5 * n).long().clamp_(max=n - 1)
    r = torch.randint(0, nr, (t,), device=dev, generator=gen)
    src, dst, rel = torch.cat([s, o]), torch.cat([o, s]), torch.cat([r, r + nr])
    order = torch.sort((dst * n + src) * (2 * nr) + rel)[1]
    trip = torch.stack([s, r, o], 1)
    return src[order], dst[order], rel[order], trip


def make_workload(rank, world, args, dev):
    from gcn_vae_amd import distributed as gdist
    from gcn_vae_amd import sampling
    from gcn_vae_amd.graph import KGraph
    cfg = CONFIGS[args.config]
    n, nr = cfg['nodes'], cfg['rels']
    strong = args.scaling == 'strong' and world > 1
    src, dst, rel, trip = directed_graph(cfg, 0 if strong else rank, dev)
    e_total = int(src.numel())
    deg = torch.bincount(dst.to(dev), minlength=n).to(torch.float32)
    shard = None
    if strong:      # ONE graph, its directed edges cut by relation range (whole relations): the (dst, src, rel) order survives
        ids, shard = gdist.shard_edges_by_relation(rel.cpu().numpy(), 2 * nr, world, rank)
        ids = torch.from_numpy(ids).to(src.device)
        src, dst, rel = src[ids], dst[ids], rel[ids]
    elif world > 1:      # 1/in-degree over the union of all ranks' edge blocks
        import torch.distributed as dist
        dist.all_reduce(deg)
    norm = torch.where(deg > 0, 1.0 / deg.clamp(min=1), torch.zeros_like(deg))
    enorm = norm[dst.to(dev)].view(-1, 1).contiguous()
    if cfg['device_gen']:
        g = KGraph.from_device_edges(n, src.to(dev), dst.to(dev), dst_sorted=True)
    else:
        g = KGraph()
        g.add_nodes(n)
        g.add_edges(src, dst)
    rs = np.random.RandomState(1000 + (0 if strong else rank))
    trip_np = trip.cpu().numpy() if isinstance(trip, torch.Tensor) else trip
    pos = trip_np[rs.permutation(len(trip_np))[:args.positives]]
    np.random.seed(7 + (0 if strong else rank))
    samples, labels = sampling.negative_sampling(pos, n, args.negative_sample)
    if strong:      # the ONE triplet batch dealt round-robin
        samples, labels = samples[rank::world], labels[rank::world]
    return dict(data=_Data(n, nr, trip_np), g=g, src=src, dst=dst, rel=rel.cpu() if not cfg['device_gen'] else rel,
                enorm=enorm, node_id=torch.arange(n, dtype=torch.long).view(-1, 1), samples=torch.from_numpy(samples),
                labels=torch.from_numpy(labels), e_total=e_total, shard=shard, strong=strong)


def make_workload_rows(rank, world, args, dev, own):
    """Destination-row partition of the SAME trained graph (weak: the union of all ranks' edge blocks, seeds 0..world-1;
    strong: the one graph): cut by destination row; this rank keeps the edges that end in its rows.  Triplets are the
    rank's own (as in the edge-block workload ``own``), relabelled to row positions."""
    from gcn_vae_amd import distributed as gdist
    cfg = CONFIGS[args.config]
    n = cfg['nodes']
    src, dst, rel = [], [], []
    for r in ([0] if own['strong'] else range(world)):   # every rank regenerates all blocks (cheap) instead of exchanging them
        s_, d_, r_, _ = directed_graph(cfg, r, dev)
        src.append(s_); dst.append(d_); rel.append(r_)
    src, dst, rel = (torch.cat(x).to(dev) for x in (src, dst, rel))
    deg = torch.bincount(dst, minlength=n)
    norm = torch.where(deg > 0, 1.0 / deg.clamp(min=1).float(), torch.zeros(n, device=dev))
    part = gdist.make_row_partition(deg.cpu().numpy(), world, rank)
    g, etype, enorm = gdist.build_row_block(part, part.pos_of_node, src, dst, rel, norm, dev)
    pos = torch.from_numpy(part.pos_of_node).to(dev)
    trip = own['samples'].to(dev)
    samples = torch.stack([pos[trip[:, 0]], trip[:, 1], pos[trip[:, 2]]], 1).contiguous()
    node_id = torch.from_numpy(np.maximum(part.node_of_pos, 0)).to(dev).view(-1, 1)
    return dict(part=part, g=g, rel=etype, enorm=enorm, node_id=node_id, samples=samples, labels=own['labels'].to(dev),
                edges=int(g.number_of_edges()), union_edges=int(src.numel()), pos_of_node=part.pos_of_node)


def build_model(w, args):
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.train import LinkPredict
    d = w['data']
    torch.manual_seed(0)
    return LinkPredict(KGVAE, d.num_nodes, args.hidden, d.num_rels, num_bases=args.n_bases, num_hidden_layers=2,
                       dropout=args.dropout, use_cuda=True, reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10,
                       n_flows=args.n_flows)


def algorithmic_bytes(tag, E, N, R, T):
    """bytes one launch must move (fp32, int32 indices), by kernel tag."""
    kind, rest = tag.split('_', 1)
    if kind == 'agg':
        tr, blk, nb = rest.split('_')
        p, q = (int(x) for x in blk.split('x'))
        nb = int(nb[2:])
        fin, fout = nb * p, nb * q
        if p == 1 and q == 1:           # DistMult backward over 2T incidences, N entity rows, T/11.. relations
            return 2 * T * (fin * 4 + 16) + N * (fout * 4 + 4) + R // 2 * fin * 4
        return E * (fin * 4 + 12) + N * (fout * 4 + 4) + R * fin * fout // nb * 4
    if kind == 'gradw':
        blk, nb = rest.split('_')
        p, q = (int(x) for x in blk.split('x'))
        nb = int(nb[2:])
        fin, fout = nb * p, nb * q
        if p == 1 and q == 1:
            return T * (2 * fin * 4 + 16) + R // 2 * fin * 4
        return E * (fin * 4 + fout * 4 + 12) + R * fin * fout // nb * 4
    return 0


def pmc_traffic_for(tag, config='c2'):
    """(HBM-side bytes per launch, provenance) of the kernel INSTANCE behind a K1 tag in this configuration, from the newest
    committed rocprofv3 PMC passes of this command (profiles/round*/pmc_traffic_<config>.json: (2*FETCH_SIZE + WRITE_SIZE) KiB,
    separate --pmc runs, gfx950 correction per MI355X_MICROARCH.md).  A figure is attached only when (a) the file's
    "_meta.k1_source_sha" equals the hash of the K1 sources of THIS tree (profiles/summarize_pmc.py: the instance the tag runs is
    then the instance that was measured) and (b) exactly one kernel of the file matches the tag's family, block shape and
    orientation; the provenance names that kernel in full.  Otherwise (None, {"stale": reason})."""
    import glob
    import re
    import importlib.util
    kind, rest = tag.split('_', 1)
    if kind != 'agg':
        return None, None
    tr, blk, _ = rest.split('_')
    p, q = blk.split('x')
    # (the LDS-resident kernel's name carries no orientation -- its packing does: both directions of a square block share it)
    pat = re.compile(r'k_agg_(fast|packed|phase|stream|split)<%s, ?%s, ?%s,|k_agg_lds<%s, ?%s,' % (p, q, 'true' if tr == 'T' else 'false', p, q))
    paths = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*', f'pmc_traffic_{config}.json')))
    if not paths:
        return None, None
    path = paths[-1]
    try:
        data = json.load(open(path))
    except Exception:
        return None, None
    meta = dict(data.get('_meta', {}), file=os.path.relpath(path, ROOT))
    spec = importlib.util.spec_from_file_location('_gv_summarize_pmc', os.path.join(ROOT, 'profiles', 'summarize_pmc.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if meta.get('k1_source_sha') != mod.k1_source_sha(ROOT):
        return None, {'stale': 'the K1 sources changed since this PMC pass (or it predates the source hash)', 'file': meta['file']}
    hits = [k for k in data if k != '_meta' and pat.search(k)]
    if len(hits) != 1:
        return None, {'stale': f'{len(hits)} kernels of the file match {tag}', 'file': meta['file']}
    return data[hits[0]]['traffic_bytes'], dict(meta, kernel=hits[0])


def k1_bf16_mode(w, args, dev):
    """Does the product run both R-GCN layers' aggregations on bf16 operands for this workload (--gemm-precision bf16 and few
    enough relation types for the LDS-resident kernel)?  The oracle mirrors exactly that (oracle/bf16.py)."""
    from gcn_vae_amd import ops as _ops
    _ops.set_gemm_precision(args.gemm_precision)
    gi = w['g'].device_index(dev)
    r2 = 2 * w['data'].num_rels
    return bool(_ops.k1_bf16_applies(gi, r2, args.n_bases, args.hidden, args.hidden) and
                _ops.k1_bf16_applies(gi, r2, args.n_bases, args.hidden, 2 * args.hidden))


def cpu_baseline(w, model, args, budget_s, k1_bf16=False):
    """The CPU oracle (oracle/, a torch-CPU port of the reference's op sequence; test infrastructure, used here as the
    reported baseline and as the checker of ``parity_check``) on the same inputs.  Returns (baseline record, reference
    outputs of the last oracle step for the parity leg)."""
    from oracle import bf16 as obf16
    from oracle import kgvae as okg
    from oracle import rgcn as orgcn
    ncpu = os.cpu_count() or 1
    src, dst, rel = (t.cpu() for t in (w['src'], w['dst'], w['rel']))
    # torch's CPU ops do not scale to hundreds of threads on this op mix: pick the fastest of a few thread
    # counts on the dominant op (one layer-1 message pass) and report the count actually used.
    xs = model.state_dict()
    probe = {'weight': xs['encoder.rconv_layer_1.weight'].detach().cpu(), 'h_bias': xs['encoder.rconv_layer_1.h_bias'].detach().cpu(),
             'loop_weight': xs['encoder.rconv_layer_1.loop_weight'].detach().cpu()}
    xprobe = xs['encoder.input_layer.embedding.weight'].detach().cpu()
    enorm = w['enorm'].cpu()
    best_t, threads = None, 1
    for cand in [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(cand)
        with torch.no_grad():
            orgcn.rel_graph_conv(xprobe, src, dst, rel, enorm, probe, 'bdd', args.n_bases, torch.relu)
            t0 = time.time()
            orgcn.rel_graph_conv(xprobe, src, dst, rel, enorm, probe, 'bdd', args.n_bases, torch.relu)
            dt = time.time() - t0
        if best_t is None or dt < best_t:
            best_t, threads = dt, cand
    torch.set_num_threads(threads)
    state = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and 'mask' not in k and not k.endswith('.pi'))
             for k, v in model.state_dict().items()}
    n, h = int(w['node_id'].shape[0]), args.hidden
    gen = torch.Generator().manual_seed(0)
    eps, eps_prior = torch.randn(n, h, generator=gen), torch.randn(200, h, generator=gen)
    keep1 = (torch.rand(n, h, generator=gen) > args.dropout).to(torch.uint8)
    keep2 = (torch.rand(n, 2 * h, generator=gen) > args.dropout).to(torch.uint8)
    post_idx = torch.tensor(random.Random(0).sample(range(n), 200))

    def one_step(anomaly=False):
        for v in state.values():
            v.grad = None
        t0 = time.time()
        # configs[2]: the oracle emulates the product's bf16-operand / fp32-accumulate dense products (oracle/bf16.py)
        with torch.autograd.set_detect_anomaly(anomaly), obf16.enabled(args.gemm_precision == 'bf16', k1=k1_bf16):
            enc = okg.kgvae_encode(state, src, dst, w['node_id'], rel, enorm, eps, args.n_bases, args.n_flows, args.dropout,
                                   keep1, keep2)
            loss = okg.link_predict_loss(state, enc, w['samples'], w['labels'], 0.01, 1e-5, 1.0, 10, args.n_flows,
                                         eps_prior, post_idx)[0]
            loss.backward()
        return time.time() - t0, enc, loss

    # SURVEY 8(d) asks for >= 5 warm-up + >= 20 timed steps; a step costs seconds here: up to 3 warm-up steps, then as many timed
    # steps (at most 20) as --cpu-seconds allows, at least 2 (the default budget covers 3 + 20 steps of the default workload)
    times, t_start = [], time.time()
    dt, enc, loss = one_step()
    warmups = 1
    while warmups < 3 and (warmups + 3) * dt < budget_s:      # up to 3 warm-up steps where the budget still leaves >= 3 timed ones
        dt, enc, loss = one_step()
        warmups += 1
    anomaly_dt = None
    if 3 * dt < budget_s + 15:
        try:
            anomaly_dt = one_step(anomaly=True)[0]
        except RuntimeError as exc:       # the reference's global anomaly mode turns a NaN in ANY backward function into an exception
            print(f'[bench] oracle step under set_detect_anomaly(True) raised: {str(exc).splitlines()[0]}', file=sys.stderr)
    while len(times) < 20 and (len(times) < 2 or time.time() - t_start + dt < budget_s):
        dt, enc, loss = one_step()
        times.append(dt)
    E = int(src.numel())
    med = float(np.median(times))
    rec = {'value': E / med, 'unit': 'edges/s', 'cores': threads, 'kind': 'port', 'steps': len(times), 'warmup_steps': warmups,
           'sample': f'{len(times)} timed full steps (fwd+loss+bwd of the same workload; clip+Adam, which the GPU span includes, '
                     f'are NOT in this span) after {warmups} warm-up, bounded by --cpu-seconds {budget_s:g} (SURVEY 8(d)\'s 5 + 20 steps '
                     f'would take minutes); median {med:.3f} s/step; torch {torch.__version__} CPU with {threads} of {ncpu} host '
                     f'threads (fastest of 8..128 on a probe), anomaly mode off'
                     + (f'; one step with the reference\'s torch.autograd.set_detect_anomaly(True) (kgvae/model.py:10): '
                        f'{anomaly_dt:.3f} s' if anomaly_dt is not None else '')}
    names = ['encoder.rconv_layer_1.weight', 'encoder.rconv_layer_2.weight', 'encoder.rconv_layer_2.loop_weight',
             'encoder.input_layer.embedding.weight', 'w_relation', 'encoder.z_pre']
    ref = dict(loss=loss.detach(), z=enc['z'].detach(), h1=enc['h1'].detach(), grads={k: state[k].grad.detach().clone() for k in names},
               eps=eps, eps_prior=eps_prior, keep1=keep1, keep2=keep2, post_idx=post_idx,
               state={k: v.detach() for k, v in state.items()})
    return rec, ref


def parity_check(model, opt, inputs, ref, dev, bf16_products=False):
    """ONE eager HIP step (forward + loss + backward, no optimiser step) on the timed workload with the random draws the
    oracle used, compared with the oracle's step on the same weights: loss, z and six parameter gradients.
    Returns the JSON record; raises AssertionError beyond north_star's tolerance (1e-4 on outputs, 5e-4 on gradients,
    relative to the tensor's largest magnitude).  ``bf16_products`` (configs[2]): the dense products take bf16 operands on
    both sides; a last-bit difference in an fp32 activation can flip its bf16 rounding (2^-9 relative on that operand),
    so the bounds are 5e-3 / 2e-2 there, as in tests/test_gpu_model.py::test_c3_wn18rr_shape_bf16_operand_gemms."""
    tol_out, tol_grad = (5e-3, 2e-2) if bf16_products else (1e-4, 5e-4)
    enc = model.encoder
    saved = (enc.eps_override, enc.mmd_eps_override, enc.mmd_index_override, enc.rconv_layer_1.keep_mask_override,
             enc.rconv_layer_2.keep_mask_override)
    with torch.no_grad():      # the oracle ran on a snapshot of the weights: make sure the device holds the same values
        for k, v in model.state_dict().items():
            if k in ref['state'] and v.is_floating_point():
                v.copy_(ref['state'][k].to(dev))
    enc.eps_override, enc.mmd_eps_override = ref['eps'].to(dev), ref['eps_prior'].to(dev)
    enc.mmd_index_override = ref['post_idx'].to(dev)
    enc.rconv_layer_1.keep_mask_override, enc.rconv_layer_2.keep_mask_override = ref['keep1'].to(dev), ref['keep2'].to(dev)
    seen = {}
    hook = enc.rconv_layer_1.register_forward_hook(lambda _m, _i, o: seen.__setitem__('h1', o.detach()))
    try:
        opt.flat_g.zero_()
        opt._mark_fresh()
        embed = model(inputs['g'], inputs['node_id'], inputs['etype'], inputs['enorm'])
        loss = model.get_loss(inputs['g'], embed, inputs['samples'], inputs['labels'])[0]
        loss.backward()
        torch.cuda.synchronize()
        named = dict(model.named_parameters())

        def rel_err(a, b):          # largest deviation relative to the reference tensor's largest magnitude
            a, b = a.detach().double().cpu(), b.double()
            return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))

        def rel_l2(a, b):
            a, b = a.detach().double().cpu(), b.double()
            return float((a - b).norm() / b.norm().clamp(min=1e-30))

        errs = {'loss': rel_err(loss.reshape(()), ref['loss'].reshape(())), 'z': rel_err(embed, ref['z'])}
        l2 = {}
        for k, g in ref['grads'].items():
            errs['grad ' + k] = rel_err(named[k].grad, g)
            l2['grad ' + k] = rel_l2(named[k].grad, g)
        # ReLU kink: a layer-1 pre-activation within fp32 rounding of 0 may land on different sides in the two
        # implementations (different summation orders); that element's gradient is then switched on in one and off in the
        # other -- a discrete, legitimate difference that a max-norm bound on the layer-1 gradients cannot absorb.  Count
        # them; with any such element the gradients are held to the Frobenius-norm bound only.
        flips = int(((seen['h1'].cpu() > 0) != (ref['h1'] > 0)).sum())
        worst_out = max(errs['loss'], errs['z'])
        worst_grad = max(v for k, v in errs.items() if k.startswith('grad '))
        worst_l2 = max(l2.values())
        ok = worst_out <= tol_out and worst_l2 <= tol_grad and (worst_grad <= tol_grad or flips > 0)
        rec = {'parity_max_rel_err': max(worst_out, worst_l2 if flips else worst_grad), 'outputs_max_rel_err': worst_out,
               'gradients_max_rel_err': worst_grad, 'gradients_max_rel_l2_err': worst_l2, 'relu_sign_flips': flips,
               'tolerance': {'outputs': tol_out, 'gradients': tol_grad}, 'passed': ok,
               'checked': sorted(errs), 'detail': {k: float('%.3g' % v) for k, v in errs.items()},
               'detail_rel_l2': {k: float('%.3g' % v) for k, v in l2.items()},
               'against': 'oracle/ (CPU restatement) on the timed workload, same weights and random draws; errors are '
                          'max |a-b| / max |b| per tensor (rel_l2: Frobenius); gradients are held to the max-norm bound '
                          'unless a layer-1 pre-activation changed sign between the two implementations (relu_sign_flips)'}
        opt.flat_g.zero_()
        assert ok, f'bench parity check failed: {errs} (rel. Frobenius: {l2}; ReLU sign flips: {flips})'
        return rec
    finally:
        hook.remove()
        (enc.eps_override, enc.mmd_eps_override, enc.mmd_index_override, enc.rconv_layer_1.keep_mask_override,
         enc.rconv_layer_2.keep_mask_override) = saved


def k4_records(ms, _ops):
    """Pops the fused-MADE-pass tags ('madechain_*', gv_made_chain / gv_made_chain_f32) out of a KernelTimer result and grades
    each: flops of one launch (ALL layers' dense 2 m n k -- the zero tiles of the masked weights count as work although the
    kernels skip them: the figure says how fast the dense product was delivered) over its HIP-event time, against the dense MFMA
    peak of the launch's operand type (MI355X_MICROARCH.md: bf16 2.5 PFLOP/s, fp32 157 TFLOP/s)."""
    k4 = {}
    for tag in [t for t in ms if t.startswith('madechain')]:
        vals = ms.pop(tag)
        vals = vals[len(vals) // 3:] if len(vals) >= 3 else vals
        avg_ms = float(np.mean(vals))
        flops = _ops.MADE_CHAIN_FLOPS.get(tag, 0.0)
        f32 = tag.endswith('_f32')
        peak = F32_MFMA_PEAK_TFLOPS if f32 else BF16_MFMA_PEAK_TFLOPS
        k4[tag] = {'avg_us': round(avg_ms * 1e3, 2), 'launches': len(vals), 'GFLOP': round(flops / 1e9, 3), 'bound': 'mfma',
                   'operands': 'f32' if f32 else 'bf16',
                   'achieved_TFLOPs': round(flops / 1e12 / (avg_ms * 1e-3), 1) if avg_ms > 0 else None, 'peak_TFLOPs': peak}
        if k4[tag]['achieved_TFLOPs']:
            k4[tag]['frac_mfma'] = k4[tag]['frac'] = round(k4[tag]['achieved_TFLOPs'] / peak, 4)
        # the bf16 chains also against HBM: the ALGORITHMIC bytes of a launch (every operand and result of a pass once: DESIGN.md §4)
        # over the same time; the launch is graded by the bound it is closer to
        nbytes = getattr(_ops, 'MADE_CHAIN_BYTES', {}).get(tag)
        if nbytes and avg_ms > 0:
            gbs = nbytes / 1e9 / (avg_ms * 1e-3)
            k4[tag].update(algorithmic_MB=round(nbytes / 1e6, 1), achieved_GBs=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 4))
            if k4[tag]['frac_hbm'] > k4[tag].get('frac', 0.0):
                k4[tag].update(bound='hbm', frac=k4[tag]['frac_hbm'])
    return k4


def rewarm(step_fn, seconds):
    """Untimed steps for ``seconds`` of wall time (at most 500) BEFORE the contract's W warm-up steps: the CPU-oracle leg keeps the
    GPU idle for ~25 s, its clocks drop, and W = 5 steps of a 1 ms step (5 ms) do not bring them back -- the first timed region of
    the mini-batch configuration read 4.5 ms per step against 0.99 in the regions after it.  The weights are put back from the
    snapshot before every timed region, so these steps leave no trace in what is timed."""
    if seconds <= 0:
        return
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        step_fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            break


EXIT_NONFINITE_LOSS = 4
EXIT_RANK_FAILED = 5           # a rank raised outside the recoverable places: it names itself and its phase on stderr, then leaves

_PHASE = ['start']


def phase(name):
    """Name the part of the run this rank is in (printed when it fails) -- and, for tests, fail here: GV_BENCH_FAIL=<rank>/<phase prefix>[/exit]
    raises in that rank when it enters a phase whose name starts with the prefix ('/exit': the process dies without a word, as a
    rank killed from outside would)."""
    _PHASE[0] = name
    spec = os.environ.get('GV_BENCH_FAIL')
    if spec:
        parts = spec.split('/')
        if len(parts) >= 2 and parts[0] == os.environ.get('RANK', '0') and name.startswith(parts[1]):
            if len(parts) > 2 and parts[2] == 'exit':
                os._exit(9)
            raise RuntimeError(f'injected failure (GV_BENCH_FAIL={spec})')



def result_exit_code(rec):
    """Exit status of a bench run from its result line: EXIT_NONFINITE_LOSS when the last timed step's loss is NaN / inf (steps
    timed on diverged weights are not a training-throughput measurement; the reference, under its global anomaly mode, would
    have stopped at the first NaN: kgvae/model.py:10), 0 otherwise."""
    return 0 if rec.get('loss_is_finite', True) else EXIT_NONFINITE_LOSS


K4_KERNEL = {'madechain_bwd': 'k_made_chain<false,true>', 'madechain_fwd': 'k_made_chain_fwd'}


def pmc_traffic_k4(tag, config, n_rows):
    """(HBM-side bytes of one TIMED launch of a bf16 chain tag, provenance) from the newest committed PMC passes of this
    configuration (as pmc_traffic_for).  The PMC passes ran the production layout -- a launch per row block -- while the timed
    launch covers all rows: the counter figure is scaled by the number of row blocks."""
    import glob
    name = K4_KERNEL.get(tag)
    if name is None or config is None:
        return None, None
    from gcn_vae_amd import made as _made
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*', f'pmc_traffic_{config}.json')))):
        try:
            data = json.load(open(path))
        except Exception:
            continue
        if name in data:
            blocks = len(_made._made_row_blocks(n_rows))
            return data[name]['traffic_bytes'] * blocks, dict(data.get('_meta') or {}, file=os.path.relpath(path, ROOT), kernel=name,
                                                             scaled_by_row_blocks=blocks)
    return None, None


def dominant_roofline(k1_roofline, detail, k4, config=None, n_rows=None):
    """The line's ``roofline`` object: the kernel that is dominant BY TIME among the instrumented launches of one step (HIP-event
    averages x launches).  With IAF blocks that is the fused MADE pass (K4, flops against the MFMA peak of its operand type);
    without, the K1 aggregations, graded by their WORST instance.  Returns (roofline, roofline_k1): roofline_k1 is the worst K1
    instance when K4 took the headline, else None."""
    t_k1 = sum(d['avg_us'] * d['launches'] for d in detail.values())
    t_k4 = sum(d['avg_us'] * d['launches'] for d in k4.values())
    if not k4 or t_k4 <= t_k1 or not any(d.get('frac') for d in k4.values()):
        return k1_roofline, None
    dom = max((t for t in k4 if k4[t].get('frac')), key=lambda t: k4[t]['avg_us'] * k4[t]['launches'])
    d = k4[dom]
    hbm = d.get('bound') == 'hbm'
    traffic, traffic_src = pmc_traffic_k4(dom, config, n_rows) if n_rows else (None, None)
    roof = {'kernel': dom, 'bound': d.get('bound', 'mfma'), 'achieved': d['achieved_GBs'] if hbm else d['achieved_TFLOPs'],
            'peak': HBM_PEAK_GBS if hbm else d['peak_TFLOPs'], 'unit': 'GB/s' if hbm else 'TFLOP/s',
            'frac': d['frac'], 'frac_mfma': d.get('frac_mfma'), 'frac_hbm': d.get('frac_hbm'), 'algorithmic_MB': d.get('algorithmic_MB'),
            'traffic': traffic, 'traffic_source': traffic_src,
            # (an ESTIMATE where the PMC pass ran one launch per row block and the timed launch covers all rows: counter bytes x blocks)
            'traffic_is_scaled_estimate': bool(traffic_src and traffic_src.get('scaled_by_row_blocks', 1) > 1),
            'traffic_over_algorithmic': round(traffic / (d['algorithmic_MB'] * 1e6), 3) if (traffic and d.get('algorithmic_MB')) else None,
            'avg_us': d['avg_us'], 'GFLOP': d['GFLOP'], 'operands': d.get('operands'),
            'share_of_instrumented_time': round(t_k4 / max(t_k1 + t_k4, 1e-9), 3),
            'instances': {t: v.get('frac') for t, v in sorted(k4.items())}}
    return roof, k1_roofline


def self_launch(n):
    """``bench.py --gpus N`` without a launcher (WORLD_SIZE unset): start the N ranks as a CHILD ``torch.distributed.run``
    job -- one process per GPU, rendezvous on 127.0.0.1 -- relay rank 0's JSON line and return the child's exit code.
    Called before this process has made any HIP call (a process that has initialised the GPU must not be replaced or
    forked into GPU work on this pool); the parent never touches the GPU at all."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               GV_BENCH_SELF_LAUNCHED='1')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:          # rank 0 prints the one JSON line; anything else a library wrote to fd 1 goes to stderr
        text = line.strip()
        is_result = False
        if text.startswith('{'):
            try:
                is_result = isinstance(json.loads(text), dict)
            except ValueError:
                pass
        print(text, file=sys.stdout if is_result else sys.stderr, flush=True)
    return proc.wait()


def launch_check(args):
    """--launch-check: the ranks meet (process group over 127.0.0.1; RCCL when every rank has a GPU, else gloo), count each
    other with one all-reduce and rank 0 prints what the real run would claim about the job.  No kernels."""
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    n_dev = torch.cuda.device_count()        # counting devices does not initialise the GPU
    backend = os.environ.get('GV_DIST_BACKEND', 'nccl' if n_dev >= world else 'gloo')
    seen = 1
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            one = torch.ones(1, device='cuda')
        else:
            dist.init_process_group('gloo')
            one = torch.ones(1)
        dist.all_reduce(one)
        seen = int(one.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({'launch_check': True, 'n_gpus': world, 'ranks_seen': seen, 'requested_gpus': args.gpus,
                          'scaling': args.scaling, 'backend': backend if world > 1 else None,
                          'self_launched': bool(os.environ.get('GV_BENCH_SELF_LAUNCHED'))}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_minibatch(args):
    """--config mb: the reference's own training regime.  One step = device batch sampler (edge sample -> relabel -> negatives
    -> split -> (dst, src, rel)-ordered graph) + CSR / relation / triplet index builders + forward + loss + backward + clip +
    Adam, recorded once and replayed as ONE hipGraph (gcn_vae_amd.graph_step); every replay draws a fresh batch.  Single GPU."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or args.gpus > 1:
        raise SystemExit('bench.py --config mb is a single-GPU configuration (the sampled sub-graph is 20 000 edges)')
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (no CPU path); the cpu_baseline leg alone is not a benchmark')
    from gcn_vae_amd import lib
    from gcn_vae_amd import ops as _ops
    from gcn_vae_amd.data import synthetic_kg
    from gcn_vae_amd.device_sampling import DeviceSampler
    from gcn_vae_amd.encoders import KGVAE
    from gcn_vae_amd.graph_step import GraphedMiniBatchStep
    from gcn_vae_amd.optim import FlatAdam
    from gcn_vae_amd.train import LinkPredict
    cfg = CONFIGS['mb']
    _ops.set_gemm_precision(args.gemm_precision)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    lib.load()
    data = synthetic_kg(cfg['nodes'], cfg['rels'], cfg['train'], seed=0)
    k, split, neg = args.positives, 0.5, args.negative_sample
    torch.manual_seed(0)
    model = LinkPredict(KGVAE, data.num_nodes, args.hidden, data.num_rels, num_bases=args.n_bases, num_hidden_layers=2,
                        dropout=args.dropout, use_cuda=True, reg_param=0.01, kl_param=1e-5, mmd_param=1.0, k=10,
                        n_flows=args.n_flows).to(dev).train()
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, max_grad_norm=1.0)
    sm = DeviceSampler(data.train, data.num_nodes, data.num_rels, dev, seed=0)
    step = GraphedMiniBatchStep(model, opt, sm, k, split, neg)
    launch = 'eager'
    if not args.no_graph:
        step.capture(warmup=3)
        launch = 'hipgraph (sampler + index builders + step)'
    else:
        for _ in range(3):
            step()
    cpu_rec = parity_rec = None
    if not args.no_cpu_baseline:
        # one batch of the synchronising sampler, BEFORE the timed steps (weights a few updates from their initialisation): the
        # oracle's step on it (reported baseline) and the HIP step held to it
        b = sm.sample(k, split, neg)
        src, dst = b.g.edges()
        wb = dict(data=_Data(data.num_nodes, data.num_rels, data.train), g=b.g, src=src.cpu(), dst=dst.cpu(), rel=b.edge_type.cpu(),
                  enorm=b.edge_norm, node_id=b.node_id.cpu(), samples=b.samples.cpu(), labels=b.labels.cpu())
        cpu_rec, ref = cpu_baseline(wb, model, args, args.cpu_seconds)
        if not args.no_check:
            inputs = dict(g=b.g, node_id=b.node_id, etype=b.edge_type, enorm=b.edge_norm, samples=b.samples, labels=b.labels)
            parity_rec = parity_check(model, opt, inputs, ref, dev, args.gemm_precision == 'bf16')
    snap = opt.snapshot()          # every timed region starts from these weights and moments (see main())
    rewarm(step, args.rewarm_seconds if (cpu_rec is not None) else 0.0)
    for _ in range(args.warmup):
        out = step()
    regions = []
    for _rep in range(max(1, args.repeats)):
        opt.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    elapsed = float(np.median(regions))      # `value`: the MEDIAN of the --repeats regions (each EXACTLY --steps steps inside the bracket)
    final_loss = float(out[0].detach())
    E = 2 * int(k * split)                       # directed edges of a batch's message-passing graph
    T = k * (neg + 1)
    # ---- K1 launch times of the same step, eager, HIP events (launch-latency regime: reported, not a roofline claim) ------
    detail, k4 = {}, {}
    if args.profile_steps > 0:
        opt.restore(snap)
        lib.TIMER = lib.KernelTimer()
        for _ in range(args.profile_steps):
            step.eager_step()
        ms = lib.TIMER.results_ms()
        lib.TIMER = None
        k4 = k4_records(ms, _ops)
        n_rows = min(2 * k, data.num_nodes)
        for tag, vals in sorted(ms.items()):
            vals = vals[len(vals) // 3:] if len(vals) >= 3 else vals
            avg_ms = float(np.mean(vals))
            nbytes = algorithmic_bytes(tag, E, n_rows, 2 * data.num_rels, T)
            detail[tag] = {'avg_us': round(avg_ms * 1e3, 2), 'launches': len(vals), 'algorithmic_MB': round(nbytes / 1e6, 2),
                           'achieved_GBs': round(nbytes / 1e9 / (avg_ms * 1e-3), 1) if avg_ms > 0 else None}
    rg = {t: d for t, d in detail.items() if t.startswith('agg_') and not t.startswith('agg_N_1x1') and d['achieved_GBs']}
    roofline = None
    if rg:
        dom = min(rg, key=lambda t: rg[t]['achieved_GBs'])
        roofline = {'kernel': dom, 'bound': 'launch latency (24 MB of algorithmic bytes per launch: SURVEY 8(d) says report, do not '
                                            'use for the roofline)', 'achieved': rg[dom]['achieved_GBs'], 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(rg[dom]['achieved_GBs'] / HBM_PEAK_GBS, 4), 'traffic': None,
                    'avg_us': rg[dom]['avg_us'], 'algorithmic_MB': rg[dom]['algorithmic_MB']}
    out_rec = {
        'metric': 'edges/sec R-GCN forward+backward, FB15k-237 emb=200',
        'value': E * args.steps / elapsed, 'unit': 'edges/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'ms_per_step_median': float(np.median(regions)) / args.steps * 1e3, 'ms_per_step_first_region': regions[0] / args.steps * 1e3,
        'ms_per_step_repeats': [round(r / args.steps * 1e3, 5) for r in regions], 'ranks_seen': 1, 'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': 'f32' if args.gemm_precision == 'f32' else 'f32 (dense products: bf16 operands, f32 accumulate)',
        'data': 'synthetic',
        'config': {'workload': '%s: per step %d triplets sampled uniformly from %d (device sampler), relabelled, %d negatives each '
                               '=> T=%d scored triplets; split %.1f => E=%d directed edges over ~10k nodes (arrays padded to %d rows); '
                               '2-layer R-GCN-VAE bdd num_bases=%d emb_dim=%d dropout %.1f, %d IAF blocks; step = sampling + index '
                               'build + fwd + loss(BCE+reg+KL+MMD) + bwd + clip + Adam'
                               % (cfg['label'], k, len(data.train), neg, T, split, E, min(2 * k, data.num_nodes), args.n_bases,
                                  args.hidden, args.dropout, args.n_flows),
                   'baseline_config': 'configs[1], mini-batch regime', 'edges_per_gpu': E, 'trained_graph_edges': E,
                   'nodes': data.num_nodes, 'triplets_per_gpu': T, 'n_flows': args.n_flows, 'gemm_precision': args.gemm_precision,
                   'launch': launch, 'parallelism': 'single GPU'},
        'final_loss': final_loss, 'loss_is_finite': bool(np.isfinite(final_loss)), 'roofline': roofline, 'roofline_detail': detail,
        'roofline_k4': k4 or None, 'k1_GBs_per_rank': None,
        'cpu_baseline': cpu_rec, 'parity_check': parity_rec,
    }
    out_rec['roofline'], out_rec['roofline_k1'] = dominant_roofline(roofline, detail, k4)
    if parity_rec is not None:
        out_rec['parity_max_rel_err'] = parity_rec['parity_max_rel_err']
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(out_rec), flush=True)
    return result_exit_code(out_rec)


def main():
    """Every failure of a rank outside the recoverable places (a refused capture falls back to eager launches on all ranks) ends the
    rank with EXIT_RANK_FAILED after ONE stderr line naming rank and phase; the launcher (torch.distributed.run) then stops the other
    ranks, and a rank blocked in a collective with a dead peer leaves the same way once the collective times out
    (distributed.collective_timeout)."""
    try:
        return _main()
    except SystemExit:
        raise
    except BaseException as exc:      # noqa: BLE001 -- the point is to leave, loudly, whatever it was
        rank = os.environ.get('RANK', '0')
        print(f'[bench] rank {rank} failed in phase "{_PHASE[0]}": {type(exc).__name__}: {str(exc).splitlines()[0] if str(exc) else ""}',
              file=sys.stderr, flush=True)
        if os.environ.get('GV_BENCH_TRACEBACK'):
            import traceback
            traceback.print_exc()
        sys.stderr.flush()
        os._exit(EXIT_RANK_FAILED)      # (no interpreter teardown: a half-built process group can hang in its destructors)


def _main():
    args = parse()
    if args.config == 'mb':
        return run_minibatch(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))       # before any HIP call in this process
    if args.launch_check:
        return launch_check(args)
    # stdout carries ONE JSON line: libraries that print banners to fd 1 (RCCL prints its version block there when a
    # communicator is created) write to stderr instead until the result line is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    from gcn_vae_amd import distributed as gdist
    from gcn_vae_amd import lib
    rank, local_rank, world = gdist.env_world()
    if world != args.gpus:      # a launcher's WORLD_SIZE wins (the driver passes the same number to both)
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running {world} rank(s)', file=sys.stderr)
        world = max(world, 1)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (no CPU path); the cpu_baseline leg alone is not a benchmark')
    from gcn_vae_amd import ops as _ops
    _ops.set_gemm_precision(args.gemm_precision)
    dist_on = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        import torch.distributed as _d
        torch.cuda.set_device(local_rank)
        _d.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank), timeout=gdist.collective_timeout())
    # GV_DIST_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks (ranks then share
    # devices; gloo moves CUDA tensors through the host).  Never a performance configuration.
    backend = os.environ.get('GV_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if world > 1 and backend == 'nccl' and n_dev < world:
        raise RuntimeError(f'{world} ranks need {world} GPUs for RCCL (found {n_dev}); GV_DIST_BACKEND=gloo shares devices')
    dev_index = local_rank % max(n_dev, 1)
    if world > 1 and backend != 'nccl':
        torch.cuda.set_device(dev_index)
        import torch.distributed as _dg
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        _dg.init_process_group(backend, timeout=gdist.collective_timeout())
    else:
        gdist.init_process_group('nccl' if world > 1 else None)
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    lib.load()
    import torch.distributed as dist

    phase('setup: workload + model')
    w = make_workload(rank, world, args, dev)
    model = build_model(w, args).to(dev).train()
    n_nodes, E = w['data'].num_nodes, int(w['src'].numel())
    cfg = CONFIGS[args.config]
    trained_edges = w['e_total'] if w['strong'] else world * E      # directed edges of the graph one step trains on
    model.static_batch = True      # the same triplets every step: build their index once, exact and locality-ordered
    params = gdist.arena_order(model)      # last-finished-in-backward first: the arena's tail reduces under layer 1's backward
    # One GPU: the step is replayed as a hipGraph.  With RCCL collectives in the step (world > 1) the default is eager
    # launching -- measured equal to graph replay on one GPU (the step is GPU-bound: ~55 launches of 5-130 us against
    # ~15 us of host work each), and it keeps RCCL out of stream capture, which only a 1-rank group could verify here.
    use_graph = not args.no_graph and ((world == 1 and not args.force_dist) or args.graph_collectives)
    use_segments = dist_on and not use_graph and not args.no_graph and not args.no_segments
    from gcn_vae_amd.optim import FlatAdam
    sharded = dist_on and args.sharded_adam
    if sharded:     # the gradient exchange IS the optimiser step: no arena all-reduce under backward
        opt = gdist.ShardedFlatAdam(params, lr=1e-3, max_grad_norm=1.0)
    else:
        opt = FlatAdam(params, lr=1e-3, max_grad_norm=1.0)      # clip_grad_norm_(1.0) + Adam over one flat arena
    reducer = gdist.BucketedArenaReduce(opt.flat_g, opt.offsets) if (dist_on and not sharded) else None
    pick_rng = random.Random(rank)
    post_idx = torch.zeros(200, dtype=torch.long, device=dev)
    model.encoder.mmd_index_override = post_idx          # static buffer: contents refreshed per step on the host
    # two pinned staging buffers used in turn, each rewritten only after the H2D copy that last read it has completed
    pinned = [torch.zeros(200, dtype=torch.long).pin_memory() for _ in range(2)]
    pinned_evt, pinned_turn = [None, None], [0]
    one = torch.ones((), device=dev)        # d(loss)/d(loss): handed to backward instead of a ones_like fill per step

    # ---- the two multi-GPU schemes.  Each mode = inputs + how the exchange is wired into the model ----------
    modes = {}
    edge_in = dict(g=w['g'], node_id=w['node_id'].to(dev), etype=w['rel'].to(dev), enorm=w['enorm'],
                   samples=w['samples'].to(dev), labels=w['labels'].to(dev), pick_map=None, seed=0)
    want = args.partition if dist_on else 'edge'
    if want in ('edge', 'auto') or not dist_on:
        modes['edge'] = edge_in
    if dist_on and want in ('row', 'auto'):
        try:
            wr = make_workload_rows(rank, world, args, dev, w)
            modes['row'] = dict(g=wr['g'], node_id=wr['node_id'], etype=wr['rel'], enorm=wr['enorm'], samples=wr['samples'],
                                labels=wr['labels'], pick_map=wr['pos_of_node'], part=wr['part'], seed=1000 + rank,
                                edges=wr['edges'])
        except Exception as exc:
            if want == 'row':
                raise
            print(f'[bench] row-partition workload failed on rank {rank}: {type(exc).__name__}: {exc}', file=sys.stderr)
            modes['row'] = None           # dropped by every rank at the agreement point below
    hook = gdist.make_reduce_hook() if dist_on else None
    cur = {}

    def configure(name):
        m = modes[name]
        enc = model.encoder
        enc.row_part = m.get('part')
        enc.rconv_layer_1.reduce_hook = enc.rconv_layer_2.reduce_hook = hook if (dist_on and name == 'edge') else None
        enc.grad_reducer = reducer if (dist_on and name == 'edge') else None      # row scheme: one piece, at the end
        if reducer is not None:
            reducer.average = name != 'row'       # row partition: every gradient is a partial sum over the ranks' rows
        if sharded:
            opt.sharded.average = name != 'row'
        # edge-block sharding replicates the node-level work: all ranks must draw the SAME dropout masks / noise;
        # the row partition draws per-rank noise for its own rows
        torch.manual_seed(m['seed'])
        cur.clear()
        cur.update(m, name=name)

    def refresh_host_inputs():
        ids = pick_rng.sample(range(n_nodes), 200)
        if cur['pick_map'] is not None:
            ids = cur['pick_map'][ids]
        k = pinned_turn[0]
        pinned_turn[0] ^= 1
        if pinned_evt[k] is not None:
            pinned_evt[k].synchronize()
        pinned[k].copy_(torch.as_tensor(ids))
        post_idx.copy_(pinned[k], non_blocking=True)
        pinned_evt[k] = torch.cuda.Event()
        pinned_evt[k].record()

    def step_body():
        opt.zero_grad()
        embed = model(cur['g'], cur['node_id'], cur['etype'], cur['enorm'])
        loss, pred, kl, mmd = model.get_loss(cur['g'], embed, cur['samples'], cur['labels'])
        loss.backward(gradient=one)
        if reducer is not None:
            reducer.finish()               # what backward could not start early, the waits, the 1/world scale
        opt.step()
        return loss

    side = torch.cuda.Stream()      # ONE side stream for warm-up and capture (autograd pins AccumulateGrad nodes to the
    #                                 stream of a parameter's first use: capturing elsewhere drags that stream into the capture)

    def warm(k):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(k):
                refresh_host_inputs()
                out = step_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return out

    # ---- per scheme: build indices + warm up eagerly (side stream, as graph capture wants), capture, and -- "auto" --
    # time a few steps of the captured program; then run the faster scheme -------------------------------------------
    def capture_current(segments):
        """(launch description, replay callable or None, static loss) for the configured scheme."""
        if segments:
            from gcn_vae_amd.segments import SegmentedGraph
            sg, out = SegmentedGraph(), None
            try:
                refresh_host_inputs()
                phase('capture: hipGraph segments')
                out = sg.capture(step_body, stream=side)
            except Exception as exc:
                print(f'[bench] segmented capture failed on rank {rank}: {type(exc).__name__}: {exc}; running eagerly',
                      file=sys.stderr)
                if os.environ.get('GV_BENCH_TRACEBACK'):
                    import traceback
                    traceback.print_exc()
                sg = None
                torch.cuda.synchronize()
            phase('capture: agreeing on the program')
            if world > 1:      # all ranks must run the same program: one failed capture sends everybody to eager launches
                ok = torch.tensor([1 if sg is not None else 0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    sg = None
            if sg is not None:
                return 'hipgraph segments (%s)' % sg.describe(), sg.replay, out
        elif use_graph:
            try:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    out = step_body()
                return 'hipgraph', gr.replay, out
            except Exception as exc:      # capture refused (e.g. a collective that cannot be captured): run eagerly
                print(f'[bench] graph capture failed on rank {rank}: {type(exc).__name__}: {exc}; running eagerly',
                      file=sys.stderr)
                torch.cuda.synchronize()
        return 'eager', None, None

    def timed_run(k, replay):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(k):
            refresh_host_inputs()
            out = replay() if replay is not None else step_body()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, out

    # candidates = scheme x launch mode.  With collectives on the step the launch mode is part of the question: segments
    # remove the per-kernel host cost but add a graph launch per segment; "auto" measures instead of guessing.
    phase('warm-up, capture and probe of the partition schemes')
    programs, probe = {}, {}
    auto = dist_on and args.partition == 'auto'
    # Every scheme / launch variant is warmed, captured and probed from the SAME initial weights and moments, and so are the timed
    # regions afterwards: the probes of `--partition auto` are dozens of updates, enough with IAF blocks to leave exp(alpha + mu) at
    # overflow (a two-rank configs[2] run then timed a NaN model and exited with status 4; an explicit --partition was fine)
    snap0 = opt.snapshot()
    def agreed(ok):
        """True only if EVERY rank got through the stage: all ranks keep or drop a scheme together."""
        if world > 1:
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1
        return ok

    for name in list(modes):
        ok = modes[name] is not None
        try:
            if ok:
                configure(name)
                warm(2)
        except Exception as exc:      # a scheme that cannot even step is dropped (by every rank), not fatal
            print(f'[bench] scheme {name!r} failed on rank {rank}: {type(exc).__name__}: {exc}', file=sys.stderr)
            ok = False
        if not agreed(ok):
            del modes[name]
            continue
        # (segments?, destination-row blocks of the edge scheme's chunked forward all-reduce)
        # (segments?, destination-row blocks of the edge scheme's chunked forward all-reduce, blocks of the row scheme's PIPELINED
        #  layer exchange -- 1: one all-gather / reduce-scatter between the layers, not overlapped with their aggregations)
        row_env = max(1, int(os.environ.get('GV_DIST_ROW_CHUNKS', '1')))
        variants = [(use_segments, 2, row_env)] if not (auto and use_segments) else [(True, 2, row_env), (False, 2, row_env)]
        if auto and use_segments and name == 'edge':
            variants.append((True, 1, row_env))        # one all-reduce per layer instead of two overlapped halves: 2 collectives fewer
        if auto and name == 'row' and world > 1:
            variants += [(use_segments, 2, rc) for rc in (2, 4) if rc != row_env]      # the pipelined exchange, 2 and 4 blocks
        for sgm, chunks, row_chunks in variants:
            opt.restore(snap0)
            _ops.DIST_FWD_CHUNKS = chunks
            rp = getattr(model.encoder, 'row_part', None)
            if rp is not None and rp.chunks != row_chunks:
                rp.chunks = row_chunks
                warm(1)        # the row blocks of this variant are cut (one host synchronisation) outside the capture
            if chunks != 2:
                warm(1)        # the row-block cut of this variant is built (one host synchronisation) outside the capture
            key = name if (len(variants) == 1 and not auto) else '%s/%s%s%s' % (name, 'segments' if sgm else 'eager',
                                                                                 '' if chunks == 2 else '/1-block',
                                                                                 '' if row_chunks == row_env else '/pipelined-%d' % row_chunks)
            programs[key] = (name,) + capture_current(sgm) + (chunks, row_chunks)
            if auto or len(modes) * len(variants) > 1:
                k = max(1, args.probe_steps)
                timed_run(1, programs[key][2])
                probe[key] = timed_run(k, programs[key][2])[0] / k * 1e3
    if not programs:
        raise RuntimeError('bench.py: no multi-GPU scheme could run a step (see the messages above)')
    chosen = min(probe, key=probe.get) if probe else next(iter(programs))   # identical on all ranks (max-reduced times)
    mode_name, launch, replay, static_loss, _ops.DIST_FWD_CHUNKS, row_chunks = programs[chosen]
    if cur.get('name') != mode_name:
        configure(mode_name)
    if getattr(model.encoder, 'row_part', None) is not None:
        model.encoder.row_part.chunks = row_chunks
    opt.restore(snap0)
    T = int(cur['samples'].shape[0])

    def run_step():
        refresh_host_inputs()
        if replay is not None:
            replay()
            return static_loss
        return step_body()

    # The CPU oracle's step on this workload (the reported baseline) and the parity leg -- the HIP step held to it -- come BEFORE
    # the timed steps: the weights are then a few updates from their initialisation.  (With IAF blocks a hundred more updates at
    # lr 1e-3 let exp(alpha + mu) overflow in fp32 on these synthetic graphs -- in the oracle exactly as in the product; the
    # reference, which switches torch's anomaly mode on globally, would stop there.)
    cpu_rec = parity_rec = None
    k1_bf = k1_bf16_mode(w, args, dev) if (rank == 0 and world == 1) else False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_rec, ref = cpu_baseline(w, model, args, args.cpu_seconds, k1_bf16=k1_bf)
        if not args.no_check:      # the timed workload, checked at its own size against the oracle (raises on failure)
            parity_rec = parity_check(model, opt, modes['edge'], ref, dev, args.gemm_precision == 'bf16')
    # Every timed region starts from the SAME weights and Adam moments (a snapshot taken here, a few updates from the
    # initialisation, put back in place before each region and before the instrumented steps): with IAF blocks a few hundred
    # updates at lr 1e-3 on these synthetic graphs let exp(alpha + mu) overflow -- in the oracle as in the product -- and a region
    # that ends on NaN weights is not a training measurement (the reference, under its global anomaly mode, would stop there:
    # kgvae/model.py:10).  The restore is outside the timed bracket; a non-finite final loss makes this program exit non-zero.
    snap = opt.snapshot()
    rewarm(run_step, args.rewarm_seconds if (cpu_rec is not None) else 0.0)
    for _ in range(args.warmup):
        loss = run_step()
    # EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks, --repeats times (every region starts
    # from the same weights): `value` is the median region, the first region's figure is reported beside it
    regions = []
    for _rep in range(max(1, args.repeats)):
        phase(f'timed region {_rep + 1}')
        opt.restore(snap)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        regions.append(time.perf_counter() - t0)
    if world > 1:
        tt = torch.tensor(regions, device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in tt.tolist()]
    elapsed = float(np.median(regions))      # `value`: the MEDIAN of the --repeats regions (each EXACTLY --steps steps inside the bracket)
    lt = loss.detach().reshape(1).clone()
    if world > 1:      # edge-block: mean of the ranks' losses; row partition: the ranks hold SHARES of the loss
        dist.all_reduce(lt)
        if cur['name'] != 'row':
            lt /= world
    final_loss = float(lt.item())
    edge_counts = None
    if world > 1 and cur['name'] == 'row':
        ec = torch.zeros(world, dtype=torch.int64, device=dev)
        ec[rank] = cur['edges']
        dist.all_reduce(ec)
        edge_counts = [int(v) for v in ec.tolist()]

    phase('roofline leg (instrumented eager steps)')
    # ---- roofline leg: the same step, eager, with HIP events around the K1 launches --------------
    roofline, detail, k4 = None, {}, {}
    per_rank_k1 = None
    if args.profile_steps > 0:
        opt.restore(snap)
        # every rank runs the instrumented steps (the collectives need all of them); each grades its OWN launches against
        # its OWN edge count, rank 0's detail goes into the line and all ranks' K1 GB/s into `k1_GBs_per_rank`
        lib.TIMER = lib.KernelTimer()
        for _ in range(args.profile_steps):
            refresh_host_inputs()
            step_body()
            if dist_on:
                torch.cuda.synchronize()
        ms = lib.TIMER.results_ms()
        lib.TIMER = None
        R = 2 * w['data'].num_rels
        E_all = E
        if cur['name'] == 'row':
            E = int(cur['edges'])
        # K4 (the fused MADE pass, gv_made_chain): an MFMA kernel -- flops of one launch over its HIP-event time, against the
        # dense bf16 MFMA peak of MI355X_MICROARCH.md.  At these sizes (1.2 GFLOP per product, 228 workgroups on 256 CUs) the
        # launch is bound by its per-layer dependency chain, not by the matrix cores: the fraction says how far.
        k4 = k4_records(ms, _ops)
        for tag, vals in sorted(ms.items()):
            vals = vals[len(vals) // 3:] if len(vals) >= 3 else vals     # drop the first (cold) third
            avg_ms = float(np.mean(vals))
            nbytes = algorithmic_bytes(tag, E, n_nodes, R, T)
            detail[tag] = {'avg_us': round(avg_ms * 1e3, 2), 'launches': len(vals),
                           'algorithmic_MB': round(nbytes / 1e6, 2),
                           'achieved_GBs': round(nbytes / 1e9 / (avg_ms * 1e-3), 1) if avg_ms > 0 else None}
        # Every K1 instance against the memory level that serves its gathers: tables beyond the 256 MiB Infinity Cache are
        # an HBM test (8 TB/s); cache-resident ones (FB15k-237: 11.6-58 MB) are graded against the guide's indexed-row
        # rates -- Infinity Cache 8.6 TB/s for the aggregations, an XCD's L2 17.8 TB/s for grad-W, whose items are laid
        # out in L2 windows -- and ALSO against 8 TB/s (frac_of_hbm_peak), the figure north_star's 40 % target is stated in
        for tag, d in detail.items():
            kind, rest = tag.split('_', 1)
            blk = rest.split('_')[1 if kind == 'agg' else 0]
            nbk = int(rest.split('_')[-1][2:])
            gathered = n_nodes * nbk * int(blk.split('x')[0]) * 4 * (2 if kind == 'gradw' else 1)
            hbm = gathered > (256 << 20)
            d['bound'] = 'hbm' if hbm else 'l2/mall'
            d['peak_GBs'] = HBM_PEAK_GBS if hbm else (L2_GATHER_GBS if kind == 'gradw' else MALL_GATHER_GBS)
            if d['achieved_GBs']:
                d['frac'] = round(d['achieved_GBs'] / d['peak_GBs'], 4)
                d['frac_of_hbm_peak'] = round(d['achieved_GBs'] / HBM_PEAK_GBS, 4)
        rg = {k: v for k, v in detail.items() if k.startswith('agg_') and not k.startswith('agg_N_1x1') and v.get('frac')}
        if rg:
            # the headline is the WORST R-GCN aggregation instance (lowest fraction of its peak), not the best one
            dom = min(rg, key=lambda k: rg[k]['frac'])
            traffic, traffic_src = pmc_traffic_for(dom, args.config if args.hidden == CONFIGS[args.config].get('hidden', args.hidden) else '-')
            roofline = {'kernel': dom, 'bound': rg[dom]['bound'], 'achieved': rg[dom]['achieved_GBs'],
                        'peak': rg[dom]['peak_GBs'], 'unit': 'GB/s', 'frac': rg[dom]['frac'],
                        'frac_of_hbm_peak': rg[dom]['frac_of_hbm_peak'], 'traffic': traffic, 'traffic_source': traffic_src,
                        'traffic_over_algorithmic': round(traffic / (rg[dom]['algorithmic_MB'] * 1e6), 3) if traffic else None,
                        'avg_us': rg[dom]['avg_us'], 'algorithmic_MB': rg[dom]['algorithmic_MB'],
                        'frac_min': min(v['frac'] for v in rg.values()), 'frac_max': max(v['frac'] for v in rg.values()),
                        'frac_of_hbm_peak_min': min(v['frac_of_hbm_peak'] for v in rg.values()),
                        'instances': {k: v['frac'] for k, v in sorted(rg.items())}}
        E = E_all
        if world > 1:      # every rank's K1 aggregation rates (GB/s of its own algorithmic bytes), keyed by instance
            mine = {k: v['achieved_GBs'] for k, v in detail.items() if k.startswith('agg_') and not k.startswith('agg_N_1x1')}
            per_rank_k1 = [None] * world
            dist.all_gather_object(per_rank_k1, mine)
    ranks_seen = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            'metric': 'edges/sec R-GCN forward+backward, FB15k-237 emb=200',
            'value': trained_edges * args.steps / elapsed, 'unit': 'edges/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'ms_per_step_median': float(np.median(regions)) / args.steps * 1e3,
            'ms_per_step_first_region': regions[0] / args.steps * 1e3,
            'ms_per_step_repeats': [round(r / args.steps * 1e3, 5) for r in regions],
            'ranks_seen': ranks_seen, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32' if args.gemm_precision == 'f32' else 'f32 (dense products: bf16 operands, f32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': '%s: %d entities, %d directed relation types, E=%d directed edges on this rank (%d in the '
                                   'trained graph), 2-layer R-GCN-VAE bdd num_bases=%d emb_dim=%d dropout %.1f, %d IAF blocks, '
                                   'DistMult decoder on T=%d triplets per rank, step = fwd + loss(BCE+reg+KL+MMD) + bwd + clip + '
                                   'Adam' % (cfg['label'], n_nodes, 2 * cfg['rels'], E, trained_edges, args.n_bases, args.hidden,
                                             args.dropout, args.n_flows, T),
                       'baseline_config': 'configs[%d]' % cfg['idx'],
                       'edges_per_gpu': E, 'trained_graph_edges': trained_edges, 'nodes': n_nodes, 'triplets_per_gpu': T,
                       'n_flows': args.n_flows, 'relation_shard': list(w['shard']) if w['shard'] else None,
                       'gemm_precision': args.gemm_precision, 'launch': launch,
                       'parallelism': ('single GPU' if not dist_on else
                                       ('ONE graph cut by relation range' if w['strong'] else 'edge-block sharding') +
                                       ' x%d, RCCL all-reduce of node embeddings' % world if cur['name'] == 'edge' else
                                       'destination-row partition x%d of %s, RCCL all-gather / reduce-scatter of node rows'
                                       % (world, 'the one graph' if w['strong'] else 'the union of the ranks\' edge blocks')),
                       'partition': cur['name'] if dist_on else None,
                       'optimizer': 'sharded clip + Adam (reduce-scatter / all-gather of the arena)' if sharded else 'replicated clip + Adam',
                       'partition_probe_ms_per_step': {k: round(v, 4) for k, v in probe.items()} or None,
                       'row_partition_edges_per_rank': edge_counts},
            'final_loss': final_loss,
            'roofline': roofline, 'roofline_detail': detail, 'roofline_k4': k4 or None, 'k1_GBs_per_rank': per_rank_k1,
        }
        std_shape = args.hidden == CONFIGS[args.config].get('hidden', args.hidden) and args.n_flows == CONFIGS[args.config].get('flows', args.n_flows)
        out['roofline'], out['roofline_k1'] = dominant_roofline(roofline, detail, k4, args.config if std_shape else None, n_nodes)
        out['cpu_baseline'], out['parity_check'] = cpu_rec, parity_rec
        out['config']['k1_operands'] = 'bf16 (fp32 accumulate, fp32 rows in memory)' if k1_bf else 'f32'
        out['loss_is_finite'] = bool(np.isfinite(final_loss))
        if parity_rec is not None:
            out['parity_max_rel_err'] = parity_rec['parity_max_rel_err']
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist_on:
        dist.destroy_process_group()
    # every rank holds the reduced loss: all of them leave with the same status
    return result_exit_code({'loss_is_finite': bool(np.isfinite(final_loss))})


if __name__ == '__main__':
    sys.exit(main() or 0)
