#!/usr/bin/env python3
"""The MADE weight-gradient product dW = g^T a as gv_gemm_bf16_nt with split-K: (200 | 400) x 200 outputs over K rows."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from gcn_vae_amd import ops
dev = torch.device('cuda:0')
for K in (65448, 184248):
    for m in (200, 400):
        a = torch.randn(m, K, device=dev).to(torch.bfloat16)
        b = torch.randn(200, K, device=dev).to(torch.bfloat16)
        c = torch.zeros(m, 200, device=dev)
        for split in (32, 64, 128, 256):
            fn = lambda: ops.gemm_bf16_nt(a, b, m, 200, K, c_f32=c, accumulate=True, split_k=split)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f'K={K} m={m} split={split}: {e0.elapsed_time(e1) * 1000 / 20:7.1f} us (GEMM + sum)')
