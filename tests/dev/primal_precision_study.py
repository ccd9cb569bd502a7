#!/usr/bin/env python3
"""Round-4 step 0 (CPU, dev tool): what does a split-precision PRIMAL pass of the ResNet couplers cost, and
through which channel -- its VALUES (s, t, tanh', the pre-activations fed forward) or the relu MASKS the tangent pass takes
from it?  Decode side of the full-size MNIST model at the REFERENCE's latents (fixture c3_mnist_stats32: 32 samples, reference
log-det / g_ij in float32 and float64), the hidden 3x3 primal convolutions emulated per scheme, tangent convolutions exact fp32.

  python tests/dev/primal_precision_study.py [n_samples]

Schemes (values / masks): a primal chain per arithmetic is carried side by side through every coupler network;
  f32/f32        the oracle as it is
  bf16x3/bf16x3  three bf16 products (hi hi + hi lo + lo hi), what KernelConfig(primal="bf16x3") runs
  bf16x3/f32     split values, fp32 masks         f32/bf16x3   the reverse
  f16x3/f16x3    three fp16 products, lo = fp16(v - hi) UNscaled, weights pre-scaled per layer by 2^k (max |w| 2^k in [2^11, 2^12))
  f16x3u/f16x3u  the same without the weight scale (lo of a 0.04-sized weight is an fp16 subnormal)
Mask flips are counted per scheme against a float64 chain of the same network on the same input.
"""
import os, sys, time, math
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_model
from oracle import cmf_oracle as O


def split2(t, dt):
    hi = t.to(dt).to(torch.float32)
    return hi, (t - hi).to(dt).to(torch.float32)


def conv_scheme(scheme):
    if scheme == "f32":
        return lambda x, w, b: F.conv2d(x, w, b, padding=1)
    dt = torch.bfloat16 if scheme.startswith("bf16") else torch.float16

    def conv(x, w, b):
        k = 0
        if scheme == "f16x3":
            k = 11 - math.floor(math.log2(float(w.abs().max())))
        xh, xl = split2(x, dt)
        wh, wl = split2(w * 2.0 ** k, dt)
        assert bool(torch.isfinite(xh).all()) and bool(torch.isfinite(wh).all())
        y = F.conv2d(xh, wh, None, padding=1) + F.conv2d(xh, wl, None, padding=1) + F.conv2d(xl, wh, None, padding=1)
        return y * 2.0 ** -k + b.view(1, -1, 1, 1)
    return conv


STATE = {"values": "f32", "masks": "f32", "flips": {}, "elems": 0}


def net_jvp_multi(sd, op, x, V):
    """oracle.net_jvp_multi for the ResNet couplers with one primal chain per arithmetic in use (+ a float64 one for the count)."""
    if op["net"] == "mlp":
        return ORIG(sd, op, x, V)
    B, d = V.shape[:2]
    p = O._net_keys(op)
    cnv = lambda t, W, pad: F.conv2d(t.reshape(B * d, *t.shape[2:]), W, None, padding=pad).reshape(B, d, W.shape[0], *t.shape[3:])
    names = sorted({STATE["values"], STATE["masks"], "f32"})
    convs = {s: conv_scheme(s) for s in names}
    n = len(op["hidden"])
    W = sd[p + "module.0.weight"]
    h0 = F.conv2d(x, W, None, padding=1)
    h = {s: h0 for s in names}
    h64 = F.conv2d(x.double(), W.double(), None, padding=1)
    hv = cnv(V, W, 1)

    def count(tag, chains, ref):
        for s in names:
            STATE["flips"][s] = STATE["flips"].get(s, 0) + int(((chains[s] > 0) != (ref > 0)).sum())
        STATE["elems"] += ref.numel()

    for i in range(1, n + 1):
        W1, b1 = sd[f"{p}module.{i}.conv1.weight"], sd[f"{p}module.{i}.conv1.bias"]
        W2, b2 = sd[f"{p}module.{i}.conv2.weight"], sd[f"{p}module.{i}.conv2.bias"]
        count("h", h, h64)
        o = {s: convs[s](torch.relu(h[s]), W1, b1) for s in names}
        o64 = F.conv2d(torch.relu(h64), W1.double(), b1.double(), padding=1)
        ov = cnv((h[STATE["masks"]] > 0).unsqueeze(1) * hv, W1, 1)
        count("o", o, o64)
        o2 = {s: convs[s](torch.relu(o[s]), W2, b2) for s in names}
        o264 = F.conv2d(torch.relu(o64), W2.double(), b2.double(), padding=1)
        ov = cnv((o[STATE["masks"]] > 0).unsqueeze(1) * ov, W2, 1)
        h = {s: o2[s] + h[s] for s in names}
        h64 = o264 + h64
        hv = ov + hv
    W, b = sd[f"{p}module.{n+2}.weight"], sd[f"{p}module.{n+2}.bias"]
    count("h", h, h64)
    o = F.conv2d(torch.relu(h[STATE["values"]]), W, b)
    ov = cnv((h[STATE["masks"]] > 0).unsqueeze(1) * hv, W, 0)
    th = torch.tanh(o)
    w = sd[p + "weights"]
    return w * th + sd[p + "bias"], w * ((1 - th ** 2).unsqueeze(1) * ov)


ORIG = O.net_jvp_multi


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.set_num_threads(os.cpu_count())
    g, meta = load_golden("c3_mnist_stats32")
    _, _, _, ops, sd = golden_model(meta)
    pre, head, flow_ops, base, prior_ops = O.split_ops(ops)
    z = g["z_low"][:n]
    ld64, off64 = g["logdet_fp64"].reshape(-1)[:n], g["l1_off_fp64"].reshape(-1)[:n]
    ld32, off32 = g["logdet"].double().reshape(-1)[:n], g["l1_off"].double().reshape(-1)[:n]
    rel = lambda a, b: ((a.double().reshape(-1) - b).abs() / b.abs())
    print(f"reference float32 vs float64, {n} samples: logdet max {rel(ld32, ld64).max():.1e} median {rel(ld32, ld64).median():.1e}   "
          f"g_ij max {rel(off32, off64).max():.1e} median {rel(off32, off64).median():.1e}", flush=True)
    O.net_jvp_multi = net_jvp_multi
    try:
        for values, masks in (("f32", "f32"), ("bf16x3", "bf16x3"), ("bf16x3", "f32"), ("f32", "bf16x3"), ("f16x3", "f16x3"), ("f16x3u", "f16x3u")):
            STATE.update(values=values, masks=masks, flips={}, elems=0)
            t0 = time.time()
            outs = []
            for i in range(0, n, 8):                       # 8 samples at a time: the (B, d, 64, 28, 28) tangents are 1.6 GB each
                with torch.no_grad():
                    jtj, xh, J = O.jtj_batched(sd, flow_ops, base, z[i:i + 8])
                    logdet, jtj, _ = O.cholesky_logdet(jtj)
                    outs.append((logdet.reshape(-1), O.metric_l1(jtj, False).reshape(-1)))
            ld, off = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            e_ld, e_off = rel(ld, ld64), rel(off, off64)
            flips = ", ".join(f"{s} {c}" for s, c in sorted(STATE["flips"].items()))
            print(f"values {values:7s} masks {masks:7s}: logdet max {e_ld.max():.1e} median {e_ld.median():.1e}   g_ij max {e_off.max():.1e} "
                  f"median {e_off.median():.1e}   mask flips vs float64 chain of {STATE['elems']:.2e}: {flips}   ({time.time() - t0:.0f} s)", flush=True)
    finally:
        O.net_jvp_multi = ORIG


if __name__ == "__main__":
    main()
