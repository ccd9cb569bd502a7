#!/usr/bin/env python3
"""GPU dev probe: duration of ONE hidden primal conv launch (16 samples in the column slots) per kernel and work-item size, at the
shard sizes of the training / CIFAR legs.  python tests/dev/f16x3_item_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import _lib
if os.environ.get("CMF_DBG_LIB"):                                  # a tools/build_variant.sh library instead of the production one
    _lib.LIB_PATH = os.path.join(ROOT, "cmf_amd/csrc/_obj", os.environ["CMF_DBG_LIB"])
from cmf_amd import engine as E

C = 64
print(f"{'H x W':>9s} {'samples':>7s} {'items64':>7s} | {'f32':>8s} {'f16 64':>8s} {'f16 32':>8s} | with residual: {'f32':>8s} {'f16 64':>8s} {'f16 32':>8s}   (us per launch)")
for H, W, B in ((32, 32, 32), (16, 16, 32), (8, 8, 32), (28, 28, 64), (14, 14, 64), (28, 28, 32), (28, 28, 128), (32, 32, 64), (32, 32, 128), (14, 14, 512)):
    HW, G = H * W, B // 16
    gen = torch.Generator().manual_seed(1)
    xg = torch.randn(G * C * HW * 16, generator=gen).cuda()
    rg = torch.randn(G * C * HW * 16, generator=gen).cuda()
    wd = torch.nn.Parameter((torch.randn(C, C, 3, 3, generator=gen) / 24).cuda())
    bias = torch.randn(C, generator=gen).cuda()
    pn = (C * HW * 16, HW * 16, 16)
    rng = torch.zeros(2, device="cuda")
    E.absmax(xg, rng[0:1])
    yg = torch.empty_like(xg)
    m = E.BitMask(B, HW, C, "cuda")
    tiles = (H // 2) * (W // 14) if W % 14 == 0 else (H // 4) * (W // 8)
    row, outs = [], {}
    for res in (None, rg):
        for prec, item in (("f32", 0), ("f16x3", 64), ("f16x3", 32)):
            kw = dict(amax_in=rng[0:1], amax_out=rng[1:2], item_channels=item) if prec == "f16x3" else {}
            run = lambda: E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias, res_t=res,
                                         precision=prec, mask_out=m.data, mask_np=m.np_bytes, **kw)
            for _ in range(5):
                run()
            outs[res is None, prec, item] = yg.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100):
                run()
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 10)
    same = all(torch.equal(outs[r, "f16x3", 32], outs[r, "f16x3", 64]) for r in (True, False))
    print(f"{H:4d} x {W:2d} {B:7d} {G * tiles:7d} | {row[0]:8.1f} {row[1]:8.1f} {row[2]:8.1f} |                {row[3]:8.1f} {row[4]:8.1f} {row[5]:8.1f}" + ("" if same else "   ITEM SIZES DIFFER"), flush=True)
