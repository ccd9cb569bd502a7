#!/usr/bin/env python3
"""cmf_gram_backward on the ACTUAL Jacobian of the full-size MNIST model (ill-conditioned J^T J) against float64."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import build, find_head
from cmf_amd import engine as E
g, meta, cfg, dens = build("c3_mnist_full")
head = find_head(dens)
z = g["z_low"][:2].float().cuda()
with torch.no_grad():
    x_hat, T = head.program.decode(z, tangents=True)
    d = head.program.d
    gr = E.gram_cholesky(T, d)
    a = torch.ones(2, device="cuda")
    dT = E.gram_backward(T, gr.jtj, a)
J = T.to_dense(d).double().cpu()
G = torch.einsum("bni,bnj->bij", J, J)
want = 2 * torch.einsum("bni,bij->bnj", J, torch.linalg.inv(G))
got = dT.to_dense(d).double().cpu()
print("cond(G):", [f"{float(torch.linalg.cond(G[b])):.2e}" for b in range(2)])
print("gram_backward rel err:", float((got - want).abs().max() / want.abs().max()))
