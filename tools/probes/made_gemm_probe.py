#!/usr/bin/env python3
"""Time the four gv_gemm_bf16_nt launch kinds of one MADE pass at FB15k-237 size (n = 14541, d = 200)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gcn_vae_amd import ops

dev = torch.device('cuda:0')
n, d = 14541, 200
bf = dict(dtype=torch.bfloat16, device=dev)
a = torch.randn(n, d, device=dev).to(torch.bfloat16)
a4 = torch.randn(n, 2 * d, device=dev).to(torch.bfloat16)
w = (torch.randn(d, d, device=dev) * 0.1).to(torch.bfloat16)
w4 = (torch.randn(2 * d, d, device=dev) * 0.1).to(torch.bfloat16)
wk4 = (torch.randn(d, 2 * d, device=dev) * 0.1).to(torch.bfloat16)
bias = torch.randn(d, device=dev)
bias4 = torch.randn(2 * d, device=dev)
cb = torch.empty(n, d, **bf)
ct = torch.empty(d, (n + 7) // 8 * 8, **bf)
cf = torch.zeros(n, d, device=dev)
cf4 = torch.zeros(n, 2 * d, device=dev)
mask = torch.randn(n, d, device=dev).to(torch.bfloat16)
kinds = {
    'fwd hidden (bias relu -> bf16 + bf16^T)': lambda: ops.gemm_bf16_nt(a, w, n, d, d, bias=bias, relu=True, c_bf16=cb, c_bf16_t=ct),
    'fwd out n=400 (bias -> f32)': lambda: ops.gemm_bf16_nt(a, w4, n, 2 * d, d, bias=bias4, c_f32=cf4),
    'bwd masked (-> bf16 + bf16^T)': lambda: ops.gemm_bf16_nt(a, w, n, d, d, mask=mask, c_bf16=cb, c_bf16_t=ct),
    'bwd masked k=400': lambda: ops.gemm_bf16_nt(a4, wk4, n, d, 2 * d, mask=mask, c_bf16=cb, c_bf16_t=ct),
    'bwd final (f32 accumulate)': lambda: ops.gemm_bf16_nt(a, w, n, d, d, c_f32=cf, accumulate=True),
    'bf16 only': lambda: ops.gemm_bf16_nt(a, w, n, d, d, c_bf16=cb),
}
for name, fn in kinds.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'{name:45s} {e0.elapsed_time(e1) * 1000 / 50:8.2f} us')
