#!/usr/bin/env python3
"""GPU dev probe for cmf_conv_tangent_f16x3: which stored values the running maximum (amax_out) covers."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
H = W = 14
gen = torch.Generator().manual_seed(H * W)
B, C, HW = 32, 64, H * W
G = B // 16
x = torch.randn(B, C, H, W, generator=gen)
w = torch.randn(C, C, 3, 3, generator=gen) / 24
bias = torch.randn(C, generator=gen)
xg = E.primal_regroup(x.cuda(), True)
wd = torch.nn.Parameter(w.cuda())
pn = (C * HW * 16, HW * 16, 16)
for trial in range(3):
    rng = torch.zeros(2, device="cuda")
    E.absmax(xg, rng[0:1])
    yg = torch.empty_like(xg)
    E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias.cuda(), precision="f16x3",
                   amax_in=rng[0:1], amax_out=rng[1:2])
    torch.cuda.synchronize()
    got = E.primal_regroup(yg.view(G, -1), False).view(B, C, H, W)
    print("amax_out", float(rng[1]), "true max", float(got.max()))
    print(" by channel half:", [float(got[:, i * 32:(i + 1) * 32].max()) for i in range(2)])
    print(" by channel tile:", [round(float(got[:, i * 16:(i + 1) * 16].max()), 3) for i in range(4)])
    print(" by sample group:", [float(got[i * 16:(i + 1) * 16].max()) for i in range(2)])
    print(" by row pair:", [round(float(got[:, :, 2 * i:2 * i + 2].max()), 3) for i in range(7)])
    print(" by sample mod 4:", [round(float(got[i::4].max()), 3) for i in range(4)])
    print(" by column half:", [round(float(got[:, :, :, :7].max()), 3), round(float(got[:, :, :, 7:].max()), 3)])
