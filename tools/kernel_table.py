#!/usr/bin/env python3
"""Per-kernel rates of one C3 evaluation (B = 512, d = 64, eval elbo), timed with HIP events on the launch stream:
launches, average duration, algorithmic GB/s and TFLOP/s (SURVEY.md section 8d's per-unit figures).  The north star asks
for the achieved HBM GB/s of the coupling pass and the MFMA utilisation of J^T J: rows acl_tangent and gram_cholesky."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import make_model, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS
from cmf_amd import engine as E
cfg, schema, shape, sd, dens = make_model(torch.device("cuda"))
B = 512
gen = torch.Generator().manual_seed(1234)
x = (torch.randint(0, 256, (B, *shape), generator=gen).float()).cuda()
kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=True, likelihood_wt=1., metric_wt=1.)
with torch.no_grad():
    dens.elbo(x.clone(), **kw)
    E.TIMER = E.KernelTimer(lambda name: True)
    dens.elbo(x.clone(), **kw)
    rows = E.TIMER.by_name()
    E.TIMER = None
print(f"{'kernel':40s} {'launches':>8s} {'avg us':>10s} {'GB/s':>9s} {'% of 8 TB/s':>11s} {'TFLOP/s':>9s}")
for name, (n, ms, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} {n:8d} {1e3 * ms / n:10.1f} {by / ms / 1e6:9.0f} {100 * by / ms / 1e6 / HBM_PEAK_GBS:10.1f}% {fl / ms / 1e9:9.1f}")
g = rows["gram_cholesky"]
print(f"gram_cholesky: {100 * g[2] / g[1] / 1e9 / FP32_MFMA_PEAK_TFLOPS:.1f} % of the fp32 MFMA peak (fused with the factorisation)")
