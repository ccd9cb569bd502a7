#!/usr/bin/env python3
"""Experiment: does sweeping the hidden convs of a ResNet coupler SAMPLE GROUP BY SAMPLE GROUP keep the
inter-conv tangents in the 256 MiB Infinity Cache and buy clock under the board's power cap?

  python tools/exp_chain.py [--B 512] [--hw 28] [--blocks 8] [--groups 512,64,32,16,8]

Runs the 2 * blocks hidden 3x3 tangent convs (conv1: h -> u, conv2: u (+ h) -> h2, ping-pong) over B samples either in one launch
per conv (G = B) or group by group (G samples per launch, all convs of a group before the next group), each variant replayed from
a HIP graph so the host never limits it.  Prints ms per chain and TFLOP/s (fp32-equivalent)."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512); ap.add_argument("--hw", type=int, default=28); ap.add_argument("--blocks", type=int, default=8)
ap.add_argument("--groups", default="512,64,32,16,8"); ap.add_argument("--iters", type=int, default=3); ap.add_argument("--lib", default=None)
args = ap.parse_args()
from cmf_amd import _lib
if args.lib:
    _lib.LIB_PATH = os.path.abspath(args.lib)
from cmf_amd import engine as E
B, H, nc, ch = args.B, args.hw, 64, 64
HW = H * H
dev = "cuda"
h = torch.randn(B, HW, nc // 16, ch, 16, device=dev)
u = torch.empty_like(h); h2 = torch.empty_like(h)
masks = [E.relu_bits(torch.randn(B, ch, H, H, device=dev)) for _ in range(2)]
ws = [torch.nn.Parameter(torch.randn(ch, ch, 3, 3, device=dev) / 24) for _ in range(2)]
hd = (ch * HW * nc, 16, ch * nc); hsl = ch * 16
per = ch * HW * nc                                           # elements per sample


def chain(G):
    a, b, c = h, u, h2
    for g0 in range(0, B, G):
        off = g0 * per
        a, b, c = h, u, h2
        for k in range(args.blocks):
            for (src, dst, res, m, w) in ((a, b, None, masks[0], ws[0]), (b, c, a, masks[1], ws[1])):
                E.conv_tangent(src, off, *hd, w, 9, dst, *hd, G, ch, ch, H, H, nc, fmode=E.F_RELU_BITS,
                               f=m.data[g0:], f_np=m.np_bytes, res_t=res, x_sl=hsl, y_sl=hsl, y_off=off, res_off=off)
            a, c = c, a


fl = 2.0 * ch * ch * 9 * HW * nc * B * 2 * args.blocks
for G in [int(g) for g in args.groups.split(",")]:
    if G > B or B % G:
        continue
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(G)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chain(G)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"B={B} {H}x{H} G={G:4d} ({B // G * 2 * args.blocks} launches): {ms:8.2f} ms per chain  {fl / ms / 1e9:7.1f} TFLOP/s fp32-equivalent", flush=True)
