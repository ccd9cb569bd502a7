#!/usr/bin/env python3
"""Precision study for the hidden 3x3 tangent convolutions (CPU, dev tool): emulate operand-splitting schemes inside the
oracle on the full-size MNIST model (fixture c3_mnist_full, B = 2, d = 64) and compare log-det / likelihood / g_ij / J with
an fp64 evaluation.  This is how the bf16x3 scheme of conv_tangent_bf16x3.hip was chosen.

  python tests/dev/emulate_precision.py [scheme ...]      schemes: f32 bf16x1 bf16x2 bf16x3 f16x1 f16x2 f16x3 wino_f32 wino_bf16x3

wino_*: the same convolutions as Winograd F(2x2, 3x3) -- input and weight transforms in fp32, the sixteen channel contractions
in fp32 or as bf16x3 split products, output transform in fp32: 2.25x fewer multiplications (a next-round lever: both split
kernels sit at the power limit of the matrix pipes); this measures what it would cost in accuracy.
"""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_model
from oracle import cmf_oracle as O

real_conv2d = F.conv2d


def split(t, dt, n):
    parts, r = [], t
    for _ in range(n):
        p = r.to(dt).to(torch.float32)
        parts.append(p)
        r = r - p
    return parts


_G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
_BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
_AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def winograd_conv(x, w, nsplit):
    """3x3, padding 1, even H and W: y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A per 4x4 input tile (stride 2)."""
    N, C, H, W = x.shape
    G, BT, AT = _G.to(x.dtype), _BT.to(x.dtype), _AT.to(x.dtype)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                                   # (co, ci, 4, 4)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                     # (N, C, H/2, W/2, 4, 4)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT)
    if nsplit:
        dt = torch.bfloat16
        Vs, Us = split(V, dt, 2), split(U, dt, 2)
        M = torch.einsum("ocil,nctuil->notuil", Us[0], Vs[0]) + torch.einsum("ocil,nctuil->notuil", Us[0], Vs[1]) \
            + torch.einsum("ocil,nctuil->notuil", Us[1], Vs[0])
    else:
        M = torch.einsum("ocil,nctuil->notuil", U, V)
    Y = torch.einsum("ij,notujk,lk->notuil", AT, M, AT)                            # (N, co, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], H, W)


def make_conv(scheme):
    if scheme == "f32":
        return real_conv2d
    if scheme.startswith("wino"):
        def wconv(x, w, b=None, **kw):
            if b is not None or w.shape[1] % 32 or w.shape[-1] != 3 or x.dtype != torch.float32 or x.shape[-1] % 2 or x.shape[-2] % 2:
                return real_conv2d(x, w, b, **kw)
            return winograd_conv(x, w, scheme.endswith("bf16x3"))
        return wconv
    dt = torch.bfloat16 if scheme.startswith("bf16") else torch.float16
    n = int(scheme[-1])

    def conv(x, w, b=None, **kw):
        if b is not None or w.shape[1] % 32 or w.shape[-1] != 3 or x.dtype != torch.float32:
            return real_conv2d(x, w, b, **kw)                      # primal convs, first / last convs: untouched
        # n = 1: hi*hi ; n = 2: (x_hi + x_lo) * w_hi ; n = 3: hi*hi + hi*lo + lo*hi   (fp32 accumulation)
        xs, ws = split(x, dt, 2 if n > 1 else 1), split(w, dt, 2 if n > 2 else 1)
        y = real_conv2d(xs[0], ws[0], None, **kw)
        if n > 1:
            y = y + real_conv2d(xs[1], ws[0], None, **kw)
        if n > 2:
            y = y + real_conv2d(xs[0], ws[1], None, **kw)
        return y
    return conv


def run(scheme, ops, sd, x, noise):
    O.F.conv2d = make_conv(scheme)
    try:
        t0 = time.time()
        parts = O.elbo(sd, ops, x.clone(), add_reconstruction=True, add_offdiagonal_metric_reg=True, noise=noise, return_parts=True)
        return parts["elbo"], parts["parts"], time.time() - t0
    finally:
        O.F.conv2d = real_conv2d


def main():
    schemes = sys.argv[1:] or ["f32", "bf16x3", "f16x1", "f16x2", "bf16x2", "bf16x1"]
    g, meta = load_golden("c3_mnist_full")
    _, _, _, ops, sd = golden_model(meta)
    x, noise = g["x"], g["noise"]
    torch.set_default_dtype(torch.float64)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    e64, p64, dt = run("f32", ops, sd64, (x + noise).double() - noise.double(), noise.double())
    torch.set_default_dtype(torch.float32)
    print(f"fp64 reference: {dt:.0f} s; keys {sorted(p64)}", flush=True)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))
    for s in schemes:
        e, p, dt = run(s, ops, sd, x, noise)
        out = {k: rel(p[k], p64[k]) for k in ("logdet", "likelihood", "l1", "J", "recon") if k in p and k in p64 and torch.is_tensor(p[k])}
        print(f"{s:7s} elbo {rel(e, e64):.2e}  " + "  ".join(f"{k} {v:.2e}" for k, v in out.items()) + f"   ({dt:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
