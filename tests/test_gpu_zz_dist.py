"""The RCCL code path of bench.py as a 1-rank torch.distributed.run job (one GPU): process-group init over 127.0.0.1,
edge-shard layer branch with async all-reduces, flat-gradient averaging, hipGraph capture with collectives."""
import json
import os
import socket
import subprocess
import time
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(cmd, env, timeout):
    """Launch a torch.distributed.run job; a rendezvous hiccup on a cold box (port taken between the probe and the bind,
    store timeout while the image pages in) gets ONE retry -- an assertion failure inside the job fails both times."""
    out = None
    for attempt in range(2):
        if attempt:
            time.sleep(5)
            cmd = [str(free_port()) if (i and cmd[i - 1] == '--master-port') else c for i, c in enumerate(cmd)]
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if out.returncode == 0:
            break
    return out


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('extra', [['--partition', 'edge'], ['--no-graph', '--partition', 'edge'],
                                   ['--no-graph', '--partition', 'row'], ['--partition', 'row'], [],
                                   ['--partition', 'row', '--n-flows', '1'], ['--partition', 'edge', '--n-flows', '1'],
                                   ['--graph-collectives', '--partition', 'edge'],
                                   ['--partition', 'edge', '--n-flows', '1', '--gemm-precision', 'bf16'],
                                   ['--partition', 'edge', '--sharded-adam'], ['--partition', 'row', '--sharded-adam']])
def test_bench_single_rank_rccl(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '3',
           '--warmup', '1', '--no-cpu-baseline', '--profile-steps', '0', '--positives', '2000'] + extra
    out = run_ranks(cmd, env, 600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 1e6
    if '--no-graph' in extra:
        assert d['config']['launch'] == 'eager'
    elif '--graph-collectives' in extra:
        assert d['config']['launch'] == 'hipgraph'
    elif extra:      # collectives on the step: a chain of hipGraph segments with RCCL launched eagerly between them
        assert d['config']['launch'].startswith('hipgraph segments'), d['config']['launch']
    if 'row' in extra:       # RCCL's all_gather_into_tensor / reduce_scatter_tensor on a 1-rank group
        assert d['config']['partition'] == 'row'
    if '--sharded-adam' in extra:       # reduce-scatter of the gradient arena, update of the rank's piece, all-gather of the parameters
        assert d['config']['optimizer'].startswith('sharded') and 0.3 < d['final_loss'] < 3.0
    assert d['final_loss'] == d['final_loss']          # not NaN


@pytest.mark.parametrize('hidden', ['200', '500'])
def test_two_ranks_edge_sharded_equals_single_process(hidden):
    """world_size 2 on ONE GPU (gloo, both ranks on cuda:0): the edge-sharded HIP path -- the ONE graph's edges cut by
    relation range (distributed.shard_edges_by_relation), reduce hooks in both layers' forward and backward, averaged
    gradients -- equals the single-process HIP run on the whole graph.  hidden = 500: BASELINE configs[3]'s width
    (5x5 / 5x10 blocks)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_WORKER_HIDDEN=hidden)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'workers', 'dist_gpu_worker.py')]
    out = run_ranks(cmd, env, 600)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    assert out.stdout.count('worst rel err') == 2, out.stdout[-1500:]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """bench.py's world_size-2 flow end to end (sharded workload, eager launches with collectives, barrier-bracketed
    timing, max over ranks, one JSON line from rank 0) on ONE GPU: GV_DIST_BACKEND=gloo lets the ranks share it."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--profile-steps', '0', '--positives', '2000', '--scaling', 'weak']
    out = run_ranks(cmd, env, 900)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['ranks_seen'] == 2
    assert d['config']['launch'] == 'eager' or d['config']['launch'].startswith('hipgraph segments')
    assert d['value'] > 0 and d['final_loss'] == d['final_loss']
    # "auto" probes both multi-GPU schemes during warm-up and runs the faster one
    assert set(d['config']['partition_probe_ms_per_step']) == {'edge/segments', 'edge/eager', 'edge/segments/1-block',
                                                               'row/segments', 'row/eager', 'row/segments/pipelined-2',
                                                               'row/segments/pipelined-4'}
    assert d['config']['partition'] in ('edge', 'row')


def test_bench_two_ranks_auto_partition_with_iaf_blocks_ends_finite():
    """`--partition auto` warms, captures and probes five programs before the timed steps: dozens of updates.  With IAF blocks
    (BASELINE configs[2]: WN18RR shape, 3 blocks, bf16 products) that used to leave exp(alpha + mu) at overflow -- the timed regions
    then started from NaN weights and bench.py exited with status 4.  Every probed program and every timed region now starts from
    the initial weights and moments."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'c3', '--steps', '3',
           '--warmup', '2', '--no-cpu-baseline', '--profile-steps', '0']
    out = run_ranks(cmd, env, 900)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['loss_is_finite'] is True and d['final_loss'] == d['final_loss']
    assert len(d['config']['partition_probe_ms_per_step']) == 7


@pytest.mark.parametrize('extra', [['--scaling', 'strong', '--partition', 'edge'], ['--scaling', 'strong', '--partition', 'row'],
                                   ['--config', 'c4', '--scaling', 'strong', '--partition', 'edge'],
                                   ['--scaling', 'strong', '--partition', 'edge', '--sharded-adam'],
                                   ['--scaling', 'strong', '--partition', 'row', '--sharded-adam']])
def test_bench_two_ranks_strong_scaling_one_graph(extra):
    """bench.py --scaling strong on two ranks sharing one GPU (gloo): ONE FB15k-237-shaped graph, its directed edges cut by
    relation range (or by destination row), its triplets dealt round-robin; value counts the one graph's edges once."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--no-cpu-baseline', '--profile-steps', '0', '--positives', '2000'] + extra
    out = run_ranks(cmd, env, 900)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['config']['trained_graph_edges'] == 544230
    assert 0.3 < d['final_loss'] < 3.0, d['final_loss']
    if 'edge' in extra:
        lo, hi = d['config']['relation_shard']
        assert 0 <= lo < hi <= 474 and 0.3 * 544230 < d['config']['edges_per_gpu'] < 0.7 * 544230
    if 'c4' in extra:
        assert d['config']['baseline_config'] == 'configs[3]' and 'emb_dim=500' in d['config']['workload']


def test_bench_starts_its_own_ranks_strong_scaling_by_default():
    """`python bench.py --gpus 2` with NO launcher (what the driver's contract line may be run as): bench.py starts the two
    ranks itself as a child torch.distributed.run job before touching the GPU, relays rank 0's JSON line and its exit code.
    Defaults: ONE graph cut across the ranks (strong scaling), both schemes probed, K1 rates of every rank in the line,
    three timed regions with their median.  (gloo: the two ranks share the box's one GPU.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--profile-steps', '2', '--positives', '2000', '--probe-steps', '2']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # stdout carries the one JSON line and nothing else
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['scaling'] == 'strong'
    assert d['config']['trained_graph_edges'] == 544230
    assert len(d['ms_per_step_repeats']) == 3 and d['ms_per_step_median'] > 0
    assert abs(d['ms_per_step'] - d['ms_per_step_median']) < 1e-9 and abs(d['ms_per_step_first_region'] - d['ms_per_step_repeats'][0]) < 1e-3
    assert len(d['k1_GBs_per_rank']) == 2 and all(v and min(v.values()) > 0 for v in d['k1_GBs_per_rank'])


@pytest.mark.parametrize('fail,expect', [('1/capture: agreeing', 'rank 1 failed in phase "capture: agreeing on the program"'),
                                         ('1/capture: agreeing/exit', 'rank 0 failed in phase "capture: agreeing on the program"'),
                                         ('0/timed region 2', 'rank 0 failed in phase "timed region 2"')])
def test_a_failing_rank_ends_the_whole_job_with_a_status_and_a_named_phase(fail, expect):
    """First-contact robustness of the multi-rank bench: a rank that raises outside the recoverable places -- or dies without a word
    ('/exit': the peer then fails in the collective it was waiting in) -- ends with EXIT_RANK_FAILED after one stderr line naming
    rank and phase; the launcher stops the other rank; `bench.py --gpus 2` returns non-zero instead of hanging until the driver's
    limit.  (GV_BENCH_FAIL injects the failure; GV_DIST_TIMEOUT bounds what a collective may wait for a dead peer.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo', GV_BENCH_FAIL=fail, GV_DIST_TIMEOUT='60')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
           '--profile-steps', '0', '--positives', '2000', '--partition', 'edge', '--probe-steps', '2']
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0, out.stderr[-2000:]
    assert expect in out.stderr, out.stderr[-3000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')], 'no result line from a failed job'
    assert time.time() - t0 < 400


@pytest.mark.parametrize('partition', ['edge', 'row'])
def test_bench_two_ranks_each_partition(partition):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--profile-steps', '0', '--positives', '2000', '--partition', partition, '--scaling', 'weak']
    out = run_ranks(cmd, env, 900)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['config']['partition'] == partition and d['value'] > 0
    assert 0.3 < d['final_loss'] < 3.0, d['final_loss']         # both schemes train the same union graph from the same init
    if partition == 'row':
        ec = d['config']['row_partition_edges_per_rank']
        assert sum(ec) == 2 * d['config']['edges_per_gpu'] and max(ec) < 1.2 * min(ec)


@pytest.mark.parametrize('flows,chunks', [('0', '1'), ('2', '1'), ('0', '4')])
def test_row_partition_single_rank_equals_whole_graph_run(flows, chunks):
    """world 1: the destination-row path (rectangular index, ops.rel_graph_conv_rows, the loss shares) with local copies
    in place of the collectives equals the ordinary single-GPU path (flows = IAF blocks on the rank's rows)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'workers', 'dist_rows_worker.py')], cwd=ROOT,
                         env=dict(os.environ, WORLD_SIZE='1', GV_WORKER_FLOWS=flows, GV_DIST_ROW_CHUNKS=chunks), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    assert out.stdout.count('worst rel err') == 1, out.stdout[-1500:]


@pytest.mark.parametrize('flows,chunks', [('0', '1'), ('2', '1'), ('0', '3'), ('2', '2')])
def test_two_ranks_row_partition_equals_single_process(flows, chunks):
    """world_size 2 on ONE GPU (gloo): every rank owns a block of node rows and the edges ending in them; all-gather of
    layer-1 rows and of z, reduce-scatter of their gradients, summed parameter gradients == the single-process run.
    flows = 2: the IAF stack runs on the rank's rows, flow_log_prob is the all-reduced mean of the row sums.
    chunks > 1 (GV_DIST_ROW_CHUNKS): the PIPELINED exchange between the two layers -- layer 1 gathers its rows block by block under
    its own aggregation, layer 2's backward reduce-scatters dL/dh1 block by block under its own K1^T -- same results."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GV_WORKER_FLOWS=flows, GV_DIST_ROW_CHUNKS=chunks)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(ROOT, 'tests', 'workers', 'dist_rows_worker.py')]
    out = run_ranks(cmd, env, 600)
    assert out.returncode == 0, (out.stdout[-1500:] + '\n' + out.stderr[-2500:])
    assert out.stdout.count('worst rel err') == 2, out.stdout[-1500:]
