#!/usr/bin/env python3
"""Where does the HIP path leave the reference on the ill-conditioned fixtures?  z_low, then the Jacobian at the FIXTURE's z_low
and at the HIP path's own z_low, J^T J, log-det."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import COND, load_golden
from test_gpu_parity import build, find_head, rel
for name in COND + ["c3_mnist_full_cond"]:
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    y = g["head_input"].cuda()
    with torch.no_grad():
        z_low, low, _ = head.program.encode(y)
        line = f"{name:24s} z_low {rel(z_low, g['z_low']):.1e} (max |z| {float(g['z_low'].abs().max()):.0f})"
        xh_f, J_f = head.jacobian(g["z_low"].cuda())
        xh_o, J_o = head.jacobian(z_low)
        if "J" in g:
            line += f"  J@fixture-z {rel(J_f, g['J']):.1e}  J@own-z {rel(J_o, g['J']):.1e}"
            d = (J_o.cpu().double() - g["J"].double()).abs()
            i = int(d.argmax()); b, r, c = i // (d.shape[1] * d.shape[2]), (i // d.shape[2]) % d.shape[1], i % d.shape[2]
            line += f"  worst entry b={b} row={r} col={c}: {float(J_o[b, r, c]):.6g} vs {float(g['J'][b, r, c]):.6g}"
        line += f"  x_hat@fixture-z {rel(xh_f, g['x_hat']):.1e}"
        head.elbo(y, add_offdiagonal_metric_reg=True)
        gr = head.last_gram
        line += f"  jtj {rel(gr.jtj, g['jtj']):.1e} logdet {rel(gr.logdet.view(-1, 1), g['logdet']):.1e}"
    print(line, flush=True)
