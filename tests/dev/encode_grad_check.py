#!/usr/bin/env python3
"""Gradient of a linear functional of z_low = encode(x) w.r.t. the coupler parameters: HIP encode_train / encode_backward against
torch.autograd through the float64 oracle's encode."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_model
from test_gpu_parity import build, find_head
from oracle import cmf_oracle as O
for name in sys.argv[1:] or ["mini_mnist", "c3_mnist_full"]:
    g, meta, cfg, dens = build(name)
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    named = dict(dens.named_parameters())
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    x = g["x"][:2].double()
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    y, lj = O.prehead(pre, x, torch.zeros_like(x))
    z_low, low_elbo, _ = O.encode(sd64, flow_ops, base, prior_ops, y)
    w = torch.randn(z_low.shape, generator=torch.Generator().manual_seed(3)).double()
    want = dict(zip(keys, torch.autograd.grad((z_low * w).sum(), [sd64[k] for k in keys], allow_unused=True)))
    zl, lowe, u, ctx, pctx = head.program.encode_train(y.float().cuda())
    print(name, "z_low err", float((zl.cpu().double() - z_low.detach()).abs().max() / z_low.detach().abs().max()))
    grads = {}
    head.program.encode_backward(ctx, w.float().cuda(), grads)
    errs = {}
    for k, wv in want.items():
        if wv is not None and float(wv.abs().max()) > 0:
            errs[k] = float((grads[named[k]].cpu().double() - wv.reshape(named[k].shape)).abs().max() / wv.abs().max())
    v = np.array(list(errs.values()))
    print(f"   {len(v)} tensors: worst {v.max():.2e}, median {np.median(v):.2e}")
    by = {}
    for k, e in errs.items():
        by.setdefault(k.count("prior."), []).append(e)
    print("   per depth median:", {d: f"{np.median(e):.1e}" for d, e in sorted(by.items())})
