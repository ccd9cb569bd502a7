#!/usr/bin/env python3
"""Per coupling layer of the encode chain (full-size MNIST model, real layer inputs): encode_train_ / encode_backward_ against
torch.autograd through the float64 oracle's acl_x_to_z."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_model
from test_gpu_parity import build, find_head
from oracle import cmf_oracle as O
from cmf_amd.bijections import AffineCouplingBijection
name = sys.argv[1] if len(sys.argv) > 1 else "c3_mnist_full"
g, meta, cfg, dens = build(name)
_, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
head = find_head(dens)
named = {p: k for k, p in dens.named_parameters()}
pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
x = g["x"][:2].double()
h, _ = O.prehead(pre, x, torch.zeros_like(x))
gen = torch.Generator().manual_seed(9)
acls = [m for m in head.program.layers if isinstance(m, AffineCouplingBijection)]
ia = 0
for op in flow_ops:
    k = op["kind"]
    if k == "acl":
        bij = acls[ia]; ia += 1
        keys = [kk for kk in sd if kk.startswith(op["prefix"]) and sd[kk].is_floating_point() and kk in {v for v in named.values()}]
        sd64 = {kk: (v.clone().requires_grad_(True) if kk in keys else v) for kk, v in sd.items()}
        xin = h.detach().clone().requires_grad_(True)
        z, lj = O.acl_x_to_z(sd64, op, xin)
        dz = torch.randn(z.shape, generator=gen).double()
        if os.environ.get("SPARSE"):
            dz = dz * (torch.rand(z.shape, generator=gen) < float(os.environ["SPARSE"])).double()
        want = torch.autograd.grad((z * dz).sum(), [sd64[kk] for kk in keys] + [xin], allow_unused=True)
        hg = h.detach().float().cuda().contiguous()
        ctx = bij.encode_train_(hg)
        zerr = float((hg.cpu().double() - z.detach()).abs().max() / z.detach().abs().max())
        d = dz.float().cuda().contiguous()
        grads = {}
        bij.encode_backward_(d, ctx, grads)
        pmap = dict(dens.named_parameters())
        errs = []
        for kk, w in zip(keys, want[:-1]):
            if w is not None and float(w.abs().max()) > 0:
                errs.append(float((grads[pmap[kk]].cpu().double() - w.reshape(pmap[kk].shape)).abs().max() / w.abs().max()))
        dxerr = float((d.cpu().double() - want[-1]).abs().max() / want[-1].abs().max())
        print(f"acl {ia:2d} {op['mask_type']:12s} shape {tuple(h.shape[1:])}: z err {zerr:.1e}  dx err {dxerr:.1e}  param grads max {max(errs):.1e} median {np.median(errs):.1e}  "
              f"|h| max {float(h.abs().max()):.1f}", flush=True)
        h = z.detach()
    elif k == "flatten":
        h = h.flatten(1)
    elif k == "squeeze":
        h = O.squeeze_x_to_z(h, op["factor"])
    elif k == "split":
        h = torch.chunk(h, 2, dim=1)[0]
