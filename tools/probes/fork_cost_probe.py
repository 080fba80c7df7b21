#!/usr/bin/env python3
"""What a fork / join of a side stream costs inside a replayed hipGraph: a chain of 40 small kernels (axpby over 1 MB) with k of
them moved to a side stream between a fork and a join, us per replay."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcn_vae_amd import lib, ops
from gcn_vae_amd.lib import ptr

dev = torch.device('cuda:0')
x = torch.randn(1 << 18, device=dev); y = torch.zeros_like(x); z = torch.zeros_like(x)
side = torch.cuda.Stream()

def body(forks, side_work):
    for i in range(40):
        if forks and i % (40 // forks) == 0:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(side_work):
                    lib.call('gv_axpby', x.numel(), None, 1.0, ptr(x), 0.5, ptr(z), lib.stream())
            lib.call('gv_axpby', x.numel(), None, 1.0, ptr(x), 0.5, ptr(y), lib.stream())
            main.wait_stream(side)
        else:
            lib.call('gv_axpby', x.numel(), None, 1.0, ptr(x), 0.5, ptr(y), lib.stream())

for forks, side_work in ((0, 0), (1, 1), (4, 1), (10, 1), (4, 4), (10, 4)):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(forks, side_work)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(forks, side_work)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print(f'{forks:2d} fork/join pairs, {side_work} side kernels each: {a.elapsed_time(b) / 50 * 1e3:7.1f} us per replay ({40 + forks * side_work} kernels)', flush=True)
