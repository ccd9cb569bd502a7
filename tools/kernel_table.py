#!/usr/bin/env python3
"""Per-kernel rates of one evaluation (eval elbo of a BASELINE configuration, default C3: B = 512, d = 64), timed with HIP events
on the launch stream: launches, average duration, algorithmic GB/s and TFLOP/s (SURVEY.md section 8d's per-unit figures).  The
north star asks for the achieved HBM GB/s of the coupling pass and the MFMA utilisation of J^T J: rows acl_tangent and
gram_cholesky.    python tools/kernel_table.py [c1|c2a|c2b|c3|c5]"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import CONFIGS, make_model, synth_batch, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS
from cmf_amd import engine as E
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c3"
dataset, over, B, off, label = CONFIGS[cfgname]
cfg, schema, shape, sd, dens = make_model(torch.device("cuda"), dataset=dataset, overrides=over)
inner = dens.module.density if schema[0]["type"] == "dequantization" else dens
x = synth_batch(dataset, shape, B, 0, "cuda")
kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=off, likelihood_wt=1., metric_wt=1.)
with torch.no_grad():
    inner.elbo(x.clone(), **kw)
    with E.timing(lambda name: True) as timer:
        inner.elbo(x.clone(), **kw)
    rows = timer.by_name()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); inner.elbo(x.clone(), **kw); e1.record(); torch.cuda.synchronize()
print(f"{label}; B = {B}; one eager evaluation: {e0.elapsed_time(e1):.3f} ms; timed kernels: {sum(r[1] for r in rows.values()):.3f} ms")
print(f"{'kernel':40s} {'launches':>8s} {'avg us':>10s} {'total ms':>9s} {'GB/s':>9s} {'% of 8 TB/s':>11s} {'TFLOP/s':>9s}")
for name, (n, ms, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:40s} {n:8d} {1e3 * ms / n:10.1f} {ms:9.3f} {by / ms / 1e6:9.0f} {100 * by / ms / 1e6 / HBM_PEAK_GBS:10.1f}% {fl / ms / 1e9:9.1f}")
if "gram_cholesky" in rows:
    g = rows["gram_cholesky"]
    print(f"gram_cholesky: {100 * g[2] / g[1] / 1e9 / FP32_MFMA_PEAK_TFLOPS:.1f} % of the fp32 MFMA peak (launch pair: flag zeroing + fused kernel)")
