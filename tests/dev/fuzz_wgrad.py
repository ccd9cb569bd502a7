#!/usr/bin/env python3
"""Random-shape cross-check of the two weight-gradient kernels (split-precision, LDS-shared vs fp32 window kernel) through the
engine wrapper: odd image sizes, 64 / 128 channels, 32 .. 128 columns, both layouts, factor none / relu."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
gen = torch.Generator().manual_seed(2)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H, W = random.choice([1, 2, 3, 5, 7, 14, 16]), random.choice([1, 2, 3, 6, 14, 17, 28])
    cin, cout = random.choice([64, 128]), random.choice([64, 128])
    nc, B = random.choice([32, 64, 96, 128]), random.choice([1, 2, 3])
    layout, fmode = random.choice(["panel", "slice"]), random.choice(["none", "relu"])
    HW = H * W
    x = torch.randn(B, cin, H, W, nc, generator=gen); gy = torch.randn(B, cout, H, W, nc, generator=gen)
    prim = torch.randn(B, cin, H, W, generator=gen).cuda()
    if layout == "panel":
        dev = lambda t: t.contiguous().cuda(); st = lambda c: (c * HW * nc, HW * nc, nc); sl = lambda c: 16
    else:
        S = nc // 16
        dev = lambda t: t.reshape(B, -1, HW, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        st = lambda c: (c * HW * nc, 16, c * nc); sl = lambda c: c * 16
    xd, gd = dev(x), dev(gy)
    outs = []
    for prec in ("f32", "bf16x3"):
        dw = torch.zeros(cout, cin, 3, 3, device="cuda")
        E.conv_tangent_wgrad(xd, 0, *st(cin), gd, 0, *st(cout), dw, 9, B, cin, cout, H, W, nc,
                             fmode=E.F_RELU if fmode == "relu" else E.F_NONE, f=prim if fmode == "relu" else None, f_np=cin * HW, f_ci=HW,
                             f_px=1, x_sl=sl(cin), y_sl=sl(cout), precision=prec)
        outs.append(dw)
    torch.cuda.synchronize()
    err = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    ok = err < 5e-5 and bool(torch.isfinite(outs[1]).all())
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} H={H} W={W} cin={cin} cout={cout} nc={nc} B={B} {layout:5s} {fmode:4s} err={err:.1e}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
