#!/usr/bin/env python3
"""GPU dev probe 2: structured inputs for cmf_conv_tangent_bf16x3_fused1x1 (residual only: the conv input is zero)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from cmf_amd import engine as E
torch.set_printoptions(linewidth=220, precision=2, sci_mode=False)
H = W = 14
B, C, nc, n_out = 1, 64, 16, 2
HW = H * W
hd, hsl = (C * HW * nc, 16, C * nc), C * 16
u = torch.zeros(B * C * HW * nc, device="cuda")
w2 = torch.nn.Parameter(torch.randn(C, C, 3, 3, device="cuda") / 24)
ones = torch.ones(B, C, H, W, device="cuda")
b_on = E.relu_bits(ones)


def run(h5, wf, tag):
    """h5: (B, HW, nc/16, C, 16) residual in the slice-major layout; wf: (n_out, C)."""
    h = h5.contiguous().view(-1).cuda()
    wfp = torch.nn.Parameter(wf.view(n_out, C, 1, 1).contiguous().cuda())
    t = E.conv_tangent_fused1x1(u, *hd, w2, h, *hd, B, C, H, W, nc, b_on, wfp, b_on, hsl, hsl)
    torch.cuda.synchronize()
    n = B * n_out * HW * nc
    p0 = t.data[:n].view(B, n_out, HW, nc).cpu()
    p1 = t.data[t.plane:t.plane + n].view(B, n_out, HW, nc).cpu()
    ref = torch.einsum("oc,bpscl->bopsl", wf, h5).reshape(B, n_out, HW, nc)
    print(f"== {tag}: max err of the sum {float((p0 + p1 - ref).abs().max()):.3g}")
    for o in range(n_out):
        print(f"  o={o} pixel 0  plane0 {p0[0, o, 0].tolist()}")
        print(f"       pixel 0  plane1 {p1[0, o, 0].tolist()}")
        print(f"       pixel 17 plane0 {p0[0, o, 17].tolist()}")
    return p0, p1


shape = (B, HW, nc // 16, C, 16)
run(torch.ones(shape), torch.ones(n_out, C), "h = 1, wf = 1: expect 32 per plane")
ch = torch.arange(C).float().view(1, 1, 1, C, 1).expand(shape)
run(ch, torch.ones(n_out, C), "h = channel index: expect 496 / 1520")
col = torch.arange(16).float().view(1, 1, 1, 1, 16).expand(shape)
run(col.contiguous(), torch.ones(n_out, C), "h = column index: expect 32 * col per plane")
wsel = torch.zeros(n_out, C); wsel[0, 5] = 1; wsel[1, 40] = 1
run(ch, wsel, "wf picks channel 5 (o=0) and 40 (o=1): expect plane0[o=0] = 5, plane1[o=1] = 40")
