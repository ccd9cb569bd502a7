#!/usr/bin/env python3
"""Reference-side yardstick for the per-sample tolerance of the full-size statistics test (tests/test_gpu_parity.py,
test_parity_statistics_full_mnist_model): how far do the ORACLE's own per-sample log-det and g_ij of the full-size MNIST model
(d = 64, 10 ResNet couplers of 8 x 64 channels: 4.3 M relu pre-activations per sample) move when its input moves by ONE float32
ulp?  A pre-activation within rounding of zero flips a relu mask and moves that sample's Jacobian discontinuously; any two fp32
implementations differ by at least this much on such a sample.  CPU only (~5 min on 8 cores):  python tests/dev/one_ulp_full.py [N]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, golden_model
from oracle import cmf_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g, meta = load_golden("c3_mnist_full")
cfg, schema, x_shape, ops, sd = golden_model(meta)
gen = torch.Generator().manual_seed(2024)                      # the sample set of the GPU statistics test
x = torch.randint(0, 256, (N, *x_shape), generator=gen).float() + torch.rand(N, *x_shape, generator=gen)
up = torch.nextafter(x, torch.full_like(x, 1e9))
res = []
t0 = time.time()
with torch.no_grad():
    for i in range(0, N, 4):
        a = O.elbo(sd, ops, x[i:i + 4], add_offdiagonal_metric_reg=True, noise=torch.zeros_like(x[i:i + 4]), return_parts=True)["parts"]
        b = O.elbo(sd, ops, up[i:i + 4], add_offdiagonal_metric_reg=True, noise=torch.zeros_like(x[i:i + 4]), return_parts=True)["parts"]
        dl = ((a["logdet"] - b["logdet"]).abs() / a["logdet"].abs()).flatten()
        d1 = ((a["l1"] - b["l1"]).abs() / a["l1"].abs()).flatten()
        for j in range(dl.numel()):
            res.append((float(dl[j]), float(d1[j])))
        print(f"samples {i}..{i + 3}: log-det {dl.tolist()}  g_ij {d1.tolist()}  ({time.time() - t0:.0f} s)", flush=True)
ld = sorted(r[0] for r in res); l1 = sorted(r[1] for r in res)
print(f"ORACLE, one-ulp input perturbation, {N} full-size samples: log-det max {ld[-1]:.2e} median {ld[len(ld) // 2]:.2e};  "
      f"g_ij max {l1[-1]:.2e} median {l1[len(l1) // 2]:.2e}")
