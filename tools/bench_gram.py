#!/usr/bin/env python3
"""Time the fused Gram + Cholesky kernel: python tools/bench_gram.py [B] [N] [d]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
def run(B, N, d, iters=20):
    nc = E.ceil16(d)
    T = E.Tangent(B, N, nc, "panel", "cuda", data=torch.randn(B * N * nc, device="cuda"))
    for _ in range(3): E.gram_cholesky(T, d, max_attempts=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): E.gram_cholesky(T, d, max_attempts=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * N * d * d * B
    print(f"B={B} N={N} d={d}: {us:8.1f} us  gram {fl/us/1e6:7.2f} TFLOP/s ({fl/us/1e6/157.3*100:.1f}% of fp32 MFMA peak)  {4.0*B*N*nc/us/1e3:7.1f} GB/s")
args = [int(a) for a in sys.argv[1:]]
if args: run(*args)
else:
    run(512, 784, 64); run(512, 16, 64); run(512, 3072, 128); run(512, 16, 128); run(4096, 784, 64)
