#!/usr/bin/env python3
"""GPU dev probe: the full-size fixtures through the grouped primal path, per primal arithmetic, against the float64 oracle
(how far is the REFERENCE's own float32 result from float64 on the conditioned model?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_model
from test_gpu_parity import build, find_head, inner, rel
from oracle import cmf_oracle as O
from cmf_amd import engine as E

for name in ("c3_mnist_full", "c3_mnist_full_cond"):
    g, meta, cfg, dens = build(name)
    _, _, _, ops, sd = golden_model(meta)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        p64 = O.elbo(sd64, ops, g["x"].double(), noise=g["noise"].double(), add_offdiagonal_metric_reg=True, return_parts=True)["parts"]
    j64, l64 = p64["jtj"], p64["logdet"]
    print(name, "reference fp32 vs fp64: jtj", f"{rel(g['jtj'], j64):.2e}", "logdet", f"{rel(g['logdet'], l64):.2e}", flush=True)
    head = find_head(dens)
    for rep in (1, 64):
        x = (g["x"] + g["noise"]).repeat(rep, 1, 1, 1).cuda()
        for primal in ("f16x3", "f32"):
            head.kernels = E.KernelConfig(primal=primal)
            with torch.no_grad():
                inner(dens, True).elbo(x, add_offdiagonal_metric_reg=True)
            gr = head.last_gram
            print(f"  rep {rep:2d} primal {primal:6s}: jtj vs fp64 {rel(gr.jtj[0:2].cpu(), j64):.2e} vs fixture {rel(gr.jtj[0:2], g['jtj']):.2e}   "
                  f"logdet vs fp64 {rel(gr.logdet[0:2].view(-1, 1).cpu(), l64):.2e} vs fixture {rel(gr.logdet[0:2].view(-1, 1), g['logdet']):.2e}", flush=True)
