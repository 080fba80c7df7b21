#!/usr/bin/env python3
"""gv_made_row_fwd / _bwd inside a hipGraph: capture, replay, time (progress lines flushed: a hang shows where)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gcn_vae_amd import ops

def say(*a):
    print(*a, flush=True)

dev = torch.device('cuda:0')
widths = [200, 200, 200, 200, 200, 400]
L = len(widths) - 1
ws = [torch.randn(widths[i + 1], widths[i], device=dev) * 0.1 for i in range(L)]
bs = [torch.randn(widths[i + 1], device=dev) for i in range(L)]
outs = [torch.empty(1, widths[i + 1], device=dev) for i in range(L)]
gws = [torch.empty(widths[i + 1], widths[i], device=dev) for i in range(L)]
gbs = [torch.empty(widths[i + 1], device=dev) for i in range(L)]
gy = torch.randn(1, 400, device=dev)

def fwd():
    ops.made_row_fwd(None, [dict(w=ws[i], bias=bs[i], relu=i < L - 1, out=outs[i]) for i in range(L)])

def bwd():
    ops.made_row_bwd(gy, [dict(w=ws[i], act=outs[i] if i < L - 1 else None, inp=outs[i - 1] if i > 0 else None, gw=gws[i], gb=gbs[i])
                          for i in range(L)])

for name, fn in (('fwd', fwd), ('bwd', bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    say(name, 'eager ok')
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
    say(name, 'captured')
    g.replay()
    torch.cuda.synchronize()
    say(name, 'replayed once')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    say(f'{name}: {e0.elapsed_time(e1) * 100:.1f} us per launch')
