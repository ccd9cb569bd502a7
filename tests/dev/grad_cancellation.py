#!/usr/bin/env python3
"""How much do the decode-side and encode-side contributions to a coupler's gradient cancel on the full-size MNIST model?
(An invertible flow reconstructs x: both sides see the same network with opposite roles.)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import build, find_head
g, meta, cfg, dens = build("c3_mnist_full")
head = find_head(dens)
named = {p: k for k, p in dens.named_parameters()}
x = (g["x"][:2] + 0).float()
from oracle import cmf_oracle as O
from conftest import golden_model
_, schema, x_shape, ops, sd = golden_model(meta)
y, lj = O.prehead(O.split_ops(ops)[0], x, torch.zeros_like(x))
y = y.cuda()
elbo, st = head.train_forward(y, add_offdiagonal_metric_reg=True)
B = 2
w = torch.full((B,), -1.0 / B, device="cuda")
lam = float(head.regularization_param)
dec = {}
out = head.head_terms_backward(st["z_low"], st["x"], g_logdet=w * -0.5, g_l1off=w * -1.0, g_rec=w * -lam, grads=dec, state=st["head"])
enc = {}
dprior = head.program.prior_backward(st["pctx"], st["u"], w * 1.0, enc)
head.program.encode_backward(st["ctx"], out["dz_low"] + dprior, enc)
ratios = []
for p in dec:
    if p in enc:
        a, b = dec[p].double(), enc[p].double()
        ratios.append(float(max(a.abs().max(), b.abs().max()) / (a + b).abs().max()))
ratios = np.array(ratios)
print(f"{len(ratios)} coupler tensors: max(|decode part|, |encode part|) / |sum|: median {np.median(ratios):.1f}, max {ratios.max():.1f}, "
      f"90th percentile {np.percentile(ratios, 90):.1f}")
