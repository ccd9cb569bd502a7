#!/usr/bin/env python3
"""GPU dev probe: cmf_conv_tangent_bf16x3_fused1x1 against the two separate launches (last hidden conv, then the 1x1 conv)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
torch.manual_seed(0)
H = W = 14
B, C, nc, n_out = 2, 64, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 2
HW = H * W
hd, hsl = (C * HW * nc, 16, C * nc), C * 16
u = torch.randn(B * C * HW * nc, device="cuda")
h = torch.randn(B * C * HW * nc, device="cuda")
w2 = torch.nn.Parameter(torch.randn(C, C, 3, 3, device="cuda") / 24)
wf = torch.nn.Parameter(torch.randn(n_out, C, 1, 1, device="cuda") / 8)
c1 = torch.randn(B, C, H, W, device="cuda")
aK = torch.randn(B, C, H, W, device="cuda")
b_c1, b_aK = E.relu_bits(c1), E.relu_bits(aK)
# separate launches
h2 = torch.empty_like(h)
E.conv_tangent(u, 0, *hd, w2, 9, h2, *hd, B, C, C, H, W, nc, res_t=h, x_sl=hsl, y_sl=hsl, fmode=E.F_RELU_BITS, f=b_c1.data, f_np=b_c1.np_bytes)
yt = torch.empty(B * n_out * HW * nc, device="cuda")
E.conv_tangent(h2, 0, *hd, wf, 1, yt, n_out * HW * nc, HW * nc, nc, B, C, n_out, H, W, nc, fmode=E.F_RELU, f=aK, f_np=C * HW, f_ci=HW, f_px=1, x_sl=hsl)
want = yt.view(B, n_out, HW, nc)
got_t = E.conv_tangent_fused1x1(u, *hd, w2, h, *hd, B, C, H, W, nc, b_c1, wf, b_aK, hsl, hsl)
n = B * n_out * HW * nc
p0, p1 = got_t.data[:n].view(B, n_out, HW, nc), got_t.data[got_t.plane:got_t.plane + n].view(B, n_out, HW, nc)
got = p0 + p1
err = (got - want).abs()
print("max |want|", float(want.abs().max()), "max err", float(err.max()), "planes finite", bool(torch.isfinite(p0).all()), bool(torch.isfinite(p1).all()))
print("err by output", [float(err[:, o].max()) for o in range(n_out)])
print("err by sample", [float(err[b].max()) for b in range(B)])
print("err by column", [round(float(err[..., c].max()), 4) for c in range(nc)])
print("err by pixel row", [round(float(err.view(B, n_out, H, W, nc)[:, :, y].max()), 4) for y in range(H)])
print("err by pixel col", [round(float(err.view(B, n_out, H, W, nc)[:, :, :, x].max()), 4) for x in range(W)])
# reference in torch for the plane split
hm = (h2.view(B, HW, nc // 16, C, 16) * (aK > 0).float().permute(0, 2, 3, 1).reshape(B, HW, 1, C, 1))
for half in range(2):
    ref = torch.einsum("oc,bpscl->bopsl", wf.view(n_out, C)[:, half * 32:(half + 1) * 32], hm[:, :, :, half * 32:(half + 1) * 32]).reshape(B, n_out, HW, nc)
    pl = (p0, p1)[half]
    print("plane", half, "max err vs torch", float((pl - ref).abs().max()), "max ref", float(ref.abs().max()))
