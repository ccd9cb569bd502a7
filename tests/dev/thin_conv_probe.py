#!/usr/bin/env python3
"""GPU dev probe: a coupler's first conv on the VALU write-stream kernel vs the MFMA kernel (forced by a zero bias), us per launch
and TB/s of output written.   python tests/dev/thin_conv_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
for cin, H, W, nc, B in ((1, 28, 28, 64, 512), (2, 14, 14, 64, 512), (3, 32, 32, 128, 32), (3, 32, 32, 32, 32), (2, 14, 14, 32, 64), (1, 28, 28, 64, 64)):
    cout, HW = 64, H * W
    x = torch.randn(B * cin * HW * nc, device="cuda")
    wd = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device="cuda") / 5)
    y = torch.empty(B * cout * HW * nc, device="cuda")
    zb = torch.zeros(cout, device="cuda")
    st, sl = (cout * HW * nc, 16, cout * nc), cout * 16
    out = []
    for bias in (None, zb):
        run = lambda: E.conv_tangent(x, 0, cin * HW * nc, HW * nc, nc, wd, 9, y, *st, B, cin, cout, H, W, nc, y_sl=sl, precision="f32", bias=bias)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 50)
    gb = 4.0 * B * cout * HW * nc / 1e9
    print(f"cin {cin} {H:2d} x {W:2d} nc {nc:3d} samples {B:3d}: {gb:5.2f} GB out | thin {out[0]:7.1f} us {gb / out[0] * 1e3:5.2f} TB/s | MFMA {out[1]:7.1f} us {gb / out[1] * 1e3:5.2f} TB/s", flush=True)
