#!/usr/bin/env python3
"""gv_gemm_bf16_nt (csrc/k_made.hip) on the shapes of one MADE pass: us per launch and TFLOP/s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_vae_amd import ops
from tools.microbench import timeit

M = int(sys.argv[1]) if len(sys.argv) > 1 else 14541
bf = dict(dtype=torch.bfloat16, device='cuda')
for n, k in ((200, 200), (400, 200), (200, 400)):
    a = torch.randn(M, k, device='cuda').to(torch.bfloat16)
    b = torch.randn(n, k, device='cuda').to(torch.bfloat16)
    af = a.float()
    bias = torch.randn(n, device='cuda')
    mask = torch.randn(M, n, device='cuda').to(torch.bfloat16)
    cb = torch.empty(M, n, **bf)
    ct = torch.empty(n, (M + 7) // 8 * 8, **bf)
    cf = torch.zeros(M, n, device='cuda')
    fl = 2.0 * M * n * k
    for name, kw in (('bf16 out', dict(c_bf16=cb)), ('bf16 + transposed', dict(c_bf16=cb, c_bf16_t=ct)),
                     ('bias relu bf16 + T', dict(bias=bias, relu=True, c_bf16=cb, c_bf16_t=ct)),
                     ('masked bf16 + T', dict(mask=mask, c_bf16=cb, c_bf16_t=ct)), ('fp32 out', dict(c_f32=cf)),
                     ('fp32 accumulate', dict(c_f32=cf, accumulate=True))):
        t = timeit(lambda: ops.gemm_bf16_nt(a, b, M, n, k, **kw))
        print(f'M={M} N={n} K={k} {name:22s}: {t:7.1f} us  {fl / t / 1e6:7.1f} TF')
    t = timeit(lambda: ops.gemm_bf16_nt(af, b, M, n, k, c_bf16=cb))
    print(f'M={M} N={n} K={k} fp32 A (rounded)       : {t:7.1f} us  {fl / t / 1e6:7.1f} TF')
# weight gradient: [200][Mtot] x [200][Mtot]^T
mt = (5 * ((M + 7) // 8 * 8))
gt = torch.randn(200, mt, device='cuda').to(torch.bfloat16)
at = torch.randn(200, mt, device='cuda').to(torch.bfloat16)
gw = torch.zeros(200, 200, device='cuda')
for sk in (8, 16, 32, 64):
    t = timeit(lambda: ops.gemm_bf16_nt(gt, at, 200, 200, mt, c_f32=gw, accumulate=True, split_k=sk))
    print(f'dW 200x200 over {mt} rows, split_k {sk:2d}: {t:7.1f} us  {2.0 * 200 * 200 * mt / t / 1e6:7.1f} TF')
x = torch.randn(M, 200, device='cuda')
xb = torch.empty(M, 200, **bf)
xt = torch.empty(200, (M + 7) // 8 * 8, **bf)
print(f'cast + transpose {M}x200: {timeit(lambda: ops.cast_bf16(x, xb, xt)):6.1f} us')
