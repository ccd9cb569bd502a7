#!/usr/bin/env python3
"""Random-shape cross-check of the two tangent conv kernels (split-precision vs fp32 MFMA) through the engine wrapper:
unusual widths / heights / channel counts / column counts / batch sizes, both layouts, with and without residual."""
import os, sys, random
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
gen = torch.Generator().manual_seed(1)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    if random.random() < 0.5:
        W, H = random.choice([14, 28, 42, 56]), random.choice([2, 4, 6, 14, 28, 30])
    else:
        W, H = random.choice([8, 16, 24, 32, 40]), random.choice([4, 8, 12, 16, 32])
    cin, cout = random.choice([32, 64, 96, 128]), random.choice([32, 64, 128])
    nc, B = random.choice([16, 32, 48, 64, 128]), random.choice([1, 2, 3, 5])
    res, layout, fmode = random.random() < 0.5, random.choice(["panel", "slice"]), random.choice(["relu", "tanh", "raw", "none", "none+obits"])
    transpose = fmode.startswith("none") and random.random() < 0.5          # the reverse sweep's transposed packs
    if fmode == "none+obits":
        res = False                                                         # output masks exclude the residual input
    HW = H * W
    x = torch.randn(B, cin, H, W, nc, generator=gen); prim = torch.randn(B, cin, H, W, generator=gen)
    r = torch.randn(B, cout, H, W, nc, generator=gen) if res else None
    wshape = (cin, cout, 3, 3) if transpose else (cout, cin, 3, 3)          # transposed: the LAYER maps cout -> cin
    w = torch.nn.Parameter((torch.randn(*wshape, generator=gen) / (cin * 9) ** 0.5).cuda())
    src = {"relu": prim, "tanh": torch.tanh(prim), "raw": (prim > 0.3).float(), "none": None, "none+obits": None}[fmode]
    src = None if src is None else src.cuda()
    fm = {"relu": E.F_RELU, "tanh": E.F_TANH, "raw": E.F_RAW, "none": E.F_NONE, "none+obits": E.F_NONE}[fmode]
    oact = torch.randn(B, cout, H, W, generator=gen).cuda() if fmode == "none+obits" else None
    if layout == "panel":
        dev = lambda t: t.contiguous().cuda(); st = lambda c: (c * HW * nc, HW * nc, nc); sl = lambda c: 16
    else:
        S = nc // 16
        dev = lambda t: t.reshape(B, -1, HW, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()
        st = lambda c: (c * HW * nc, 16, c * nc); sl = lambda c: c * 16
    xd, rd = dev(x), (dev(r) if res else None)
    ys = []
    for prec in ("f32", "bf16x3"):
        y = torch.full((B * cout * HW * nc,), float("nan"), device="cuda")
        if oact is None:
            fo = {}
        elif prec == "bf16x3" and cout % 64 == 0 and E._shape_ok_bf16x3(9, cin, W, transpose, H, cout):
            fo = dict(fo=E.relu_bits(oact))                                 # bit mask on the split kernel
        else:
            fo = dict(fo=oact, fo_np=cout * HW, fo_co=HW, fo_px=1, fomode=E.F_RELU)   # float factor on the fp32 kernel
        E.conv_tangent(xd, 0, *st(cin), w, 9, y, *st(cout), B, cin, cout, H, W, nc, fmode=fm, f=src, f_np=cin * HW, f_ci=HW, f_px=1,
                       res_t=rd, x_sl=sl(cin), y_sl=sl(cout), precision=prec, transpose=transpose, **fo)
        ys.append(y)
    torch.cuda.synchronize()
    err = float((ys[0] - ys[1]).abs().max() / ys[0].abs().max())
    ok = err < 2e-5 and bool(torch.isfinite(ys[1]).all())
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} W={W} H={H} cin={cin} cout={cout} nc={nc} B={B} res={int(res)} {layout:5s} {fmode:10s} T={int(transpose)} split={E._shape_ok_bf16x3(9, cin, W, False, H, cout)} err={err:.1e}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
