#!/usr/bin/env python3
"""Micro-benchmark of the tangent-conv kernels through the C ABI (optionally against debug builds of the library).

  python tools/bench_conv.py [--lib path.so] [--B 128] [--hw 28] [--res 0|1] [--precision bf16x3|f32]
"""
import argparse, ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None); ap.add_argument("--B", type=int, default=128); ap.add_argument("--hw", type=int, default=28)
ap.add_argument("--w", type=int, default=None, help="image width when it differs from the height --hw")
ap.add_argument("--res", type=int, default=1); ap.add_argument("--precision", default="bf16x3"); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--fmode", default="relu"); ap.add_argument("--nc", type=int, default=64)
ap.add_argument("--wgrad", action="store_true", help="time cmf_conv_tangent_wgrad (weight gradient) instead of the forward conv")
ap.add_argument("--live", type=int, default=0, help="1 / 2: checkerboard output (compact), the pixels with (row + col) % 2 == live - 1")
ap.add_argument("--layout", default="slice", help="slice = [px][slice][ch][16] hidden layout, panel = [ch][px][nc]")
args = ap.parse_args()
from cmf_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from cmf_amd import engine as E
E.scope(tangent=args.precision).__enter__()               # for the whole process
B, H, nc, ch = args.B, args.hw, args.nc, 64
Wd = args.w or H
HW = H * Wd
torch.manual_seed(0)
x = torch.randn(B, ch, H, Wd, nc, device="cuda"); prim = torch.randn(B, ch, H, Wd, device="cuda")
res = torch.randn(B, ch, H, Wd, nc, device="cuda") if args.res else None
y = torch.randn(B, ch, H, Wd, nc, device="cuda") if args.wgrad else torch.empty(B, ch, H, Wd, nc, device="cuda")
w = torch.nn.Parameter(torch.randn(ch, ch, 3, 3, device="cuda") / 24)
fm = {"relu": E.F_RELU, "none": E.F_NONE, "bits": E.F_RELU}[args.fmode]
wg_f = E.relu_bits(prim) if args.fmode == "bits" else prim      # --wgrad: relu' as a bit mask instead of the float activation
st = (ch * HW * nc, 16, ch * nc) if args.layout == "slice" else (ch * HW * nc, HW * nc, nc)
sl = ch * 16 if args.layout == "slice" else 16
dw = torch.zeros(ch, ch, 3, 3, device="cuda")
def run_wgrad():
    E.conv_tangent_wgrad(x, 0, *st, y, 0, *st, dw, 9, B, ch, ch, H, Wd, nc, fmode=fm, f=wg_f if fm else None, f_np=ch * HW, f_ci=HW,
                         f_px=1, x_sl=sl, y_sl=sl)
def run():
    if args.wgrad:
        return run_wgrad()
    yst = (st[0] // 2, st[1], st[2]) if args.live else st          # compact output: half the pixels per sample
    E.conv_tangent(x, 0, *st, w, 9, y, *yst, B, ch, ch, H, Wd, nc, fmode=fm if args.fmode != "bits" else E.F_RELU_BITS,
                   f=(wg_f.data if args.fmode == "bits" else prim) if fm else None, f_np=wg_f.np_bytes if args.fmode == "bits" else ch * HW,
                   f_ci=HW, f_px=1, res_t=res, x_sl=sl, y_sl=sl, live=args.live, res_np=st[0])
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.iters
fl = 2.0 * ch * ch * 9 * HW * nc * B * (0.5 if args.live else 1.0)
chk = (dw if args.wgrad else y).double()
print(f"{os.path.basename(args.lib or 'libcmf_amd.so'):28s} B={B} {H}x{Wd} res={args.res} {'wgrad f32' if args.wgrad else args.precision} {args.layout}: {ms:7.3f} ms  {fl/ms/1e9:7.1f} TFLOP/s  "
      f"{4.0*HW*nc*B*ch*(2+args.res)/ms/1e6:7.1f} GB/s  checksum {float(chk.sum()):.9e} {float(chk.abs().sum()):.9e}")
