#!/usr/bin/env python3
"""Cotangent after every step of the encode chain's backward (full-size MNIST model): HIP vs float64 oracle autograd."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_model
from test_gpu_parity import build, find_head
from oracle import cmf_oracle as O
from cmf_amd import engine as E
from cmf_amd.bijections import AffineCouplingBijection
from cmf_amd.densities import SplitDensity
name = sys.argv[1] if len(sys.argv) > 1 else "c3_mnist_full"
g, meta, cfg, dens = build(name)
_, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
head = find_head(dens)
prog = head.program
pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
x = g["x"][:2].double()
y, _ = O.prehead(pre, x, torch.zeros_like(x))
# oracle chain with retained intermediate inputs
hs = []
h = y.clone().requires_grad_(True)
hs.append(h)
for op in flow_ops:
    k = op["kind"]
    if k == "acl":
        h, _ = O.acl_x_to_z(sd, op, h)
    elif k == "flatten":
        h = h.flatten(1)
    elif k == "squeeze":
        h = O.squeeze_x_to_z(h, op["factor"])
    elif k == "split":
        h = torch.chunk(h, 2, dim=1)[0]
    h.retain_grad()
    hs.append(h)
z_low = O.tail_gather(sd, base, h)
w = torch.randn(z_low.shape, generator=torch.Generator().manual_seed(3)).double()
(z_low * w).sum().backward()
print("flow ops:", [op["kind"] for op in flow_ops])
print("program :", [type(m).__name__[:12] for m in prog.layers])
# HIP chain
zl, lowe, u, ctx, pctx = prog.encode_train(y.float().cuda())
B, dev = 2, "cuda"
N = int(np.prod(prog.tail.x_shape))
dh = E.gather_primal(w.float().cuda().contiguous(), prog.tail.scatter_index(dev), N).view(B, *prog.tail.x_shape)
def err(a, b):
    return float((a.detach().cpu().double().reshape(b.shape) - b).abs().max() / b.abs().max())
print(f"after tail: {err(dh, hs[-1].grad):.1e}")
grads = {}
i = len(hs) - 1
for m, c in zip(reversed(prog.layers), reversed(ctx)):
    if isinstance(m, AffineCouplingBijection):
        m.encode_backward_(dh, c, grads)
    elif isinstance(m, SplitDensity):
        idx = torch.cat((torch.arange(c, dtype=torch.int32, device=dev), torch.full((c,), -1, dtype=torch.int32, device=dev)))
        dh = E.gather_primal(dh, idx, 2 * c).view(B, 2 * dh.shape[1], *dh.shape[2:])
    else:
        dh = m.decode(dh, None)[0]
    i -= 1
    print(f"after {type(m).__name__[:24]:24s} -> cotangent of input {tuple(dh.shape[1:])}: err {err(dh, hs[i].grad):.1e}", flush=True)

# --- isolate the first backward step (the last coupling layer of the encode chain) ---
last = [m for m in prog.layers if isinstance(m, AffineCouplingBijection)][-1]
dh0 = E.gather_primal(w.float().cuda().contiguous(), prog.tail.scatter_index(dev), N).view(B, *prog.tail.x_shape)
hin = hs[-2].detach().float().cuda().contiguous()
fresh = last.encode_train_(hin.clone())
d1 = dh0.clone(); last.encode_backward_(d1, fresh, {})
print(f"first step, FRESH ctx from the oracle's layer input: err {err(d1, hs[-2].grad):.1e}")
chain_ctx = ctx[-1]
d2 = dh0.clone(); last.encode_backward_(d2, chain_ctx, {})
print(f"first step, ctx kept by encode_train:               err {err(d2, hs[-2].grad):.1e}")
names = ["xb", "y", "g"]
for nm, a, b in zip(names, fresh[:3], chain_ctx[:3]):
    print(f"   {nm}: fresh vs chain max abs diff {float((a - b).abs().max()):.2e} (max |.| {float(a.abs().max()):.2e})")
for i, (a, b) in enumerate(zip(fresh[3], chain_ctx[3])):
    flips = int(((a > 0) != (b > 0)).sum())
    if flips or i in (0, len(fresh[3]) - 1):
        print(f"   act {i}: max abs diff {float((a - b).abs().max()):.2e}, relu sign flips {flips} of {a.numel()}")
