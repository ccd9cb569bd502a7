#!/usr/bin/env python3
"""Full-size MNIST model (d = 64, B = 2): parameter gradients of the HIP path against torch.autograd through the float64 oracle,
per arithmetic setting and per loss term -- separates precision effects of the split kernels from logic errors on the
full-size code paths.   python tests/dev/grad_full_check.py [f32 | bf16x3 | bf16x3+wgf32 ...]   (VERBOSE=1: per tensor)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_model
from test_gpu_parity import build, find_head
from oracle import cmf_oracle as O
from cmf_amd import engine as E
g, meta, cfg, dens = build("c3_mnist_full")
_, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
head = find_head(dens)
named = dict(dens.named_parameters())
keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
x = g["x"][:2].double()
KWS = {"full": dict(add_offdiagonal_metric_reg=True), "lik": dict(add_reconstruction=False), "rec": dict(likelihood_wt=0.),
       "lik+l1": dict(add_reconstruction=False, add_offdiagonal_metric_reg=True)}
y, lj = O.prehead(O.split_ops(ops)[0], x, torch.zeros_like(x))
for setting in sys.argv[1:] or ["f32", "bf16x3"]:
    tp, wg = (setting.split("+") + ["same"])[:2]
    head.kernels = E.KernelConfig(tangent=tp, primal=head.kernels.primal)
    for kwname, kw in KWS.items():
        want_elbo = O.elbo(sd64, ops, x, noise=torch.zeros_like(x), **kw)["elbo"]
        want = dict(zip(keys, torch.autograd.grad(-want_elbo.mean(), [sd64[k] for k in keys], allow_unused=True)))
        orig = E.conv_tangent_wgrad
        if wg == "wgf32":
            E.conv_tangent_wgrad = lambda *a, **k: orig(*a, **{**k, "precision": "f32"})
        loss, elbo, grads = head.loss_and_gradients(y.float().cuda(), pre_logjac=lj.float().reshape(-1).cuda(), **kw)
        E.conv_tangent_wgrad = orig
        errs = {}
        for k, w in want.items():
            if w is not None and float(w.abs().max()) > 0 and named[k] in grads:
                errs[k] = float((grads[named[k]].cpu().double() - w.reshape(named[k].shape)).abs().max() / w.abs().max())
        v = np.array(list(errs.values()))
        wk = max(errs, key=errs.get)
        print(f"{setting:13s} {kwname:7s} elbo err {float((elbo.detach().cpu().double() - want_elbo.detach()).abs().max() / want_elbo.detach().abs().max()):.1e}  "
              f"grads: worst {v.max():.2e} ({wk[-40:]}), median {np.median(v):.2e}, > 1e-4: {(v > 1e-4).sum()} of {len(v)}", flush=True)
        if os.environ.get("VERBOSE"):
            for k in keys:
                if k in errs:
                    print(f"   depth {k.count('prior.'):2d} {k.split('prior.')[-1][-58:]:58s} {errs[k]:.1e}  |g| {float(want[k].abs().max()):.2e}")
