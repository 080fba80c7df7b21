"""What one dependent kernel node of a replayed hipGraph costs on this box: chains of tiny launches (one stream), the same chain with
every launch preceded by a fork to / join from a side stream, and a chain of medium kernels (20 us of work) for comparison.
    python tools/probes/node_latency.py"""
import torch

x = torch.zeros(64, device='cuda')
big = torch.zeros(64 * 1024 * 1024, device='cuda')
side = torch.cuda.Stream()


def replay_time(fn, n):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(2)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(n)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


def chain(n):
    for _ in range(n):
        x.add_(1.0)


def chain_fork(n):
    cur = torch.cuda.current_stream()
    for _ in range(n):
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y = x * 2.0
        x.add_(1.0)
        cur.wait_stream(side)


def chain_big(n):
    for _ in range(n):
        big.add_(1.0)


for name, fn, n in (('tiny chain', chain, 200), ('tiny chain + fork/join of a tiny side kernel per node', chain_fork, 100), ('256 MB add chain', chain_big, 20)):
    t_n, t_2n = replay_time(fn, n), replay_time(fn, 2 * n)
    print(f'{name}: {n} nodes {t_n:.1f} us, {2 * n} nodes {t_2n:.1f} us -> {(t_2n - t_n) / n:.2f} us per added node', flush=True)
