#!/usr/bin/env python3
"""Training step (forward + backward + flat Adam) of the C3 / C4 model on one MI355X: the secondary metric of SURVEY 8d
("train-mode fwd+bwd steps/s").  The reference trains MNIST with 64 samples per GPU (C4).

  python tools/bench_train.py [--batch 64] [--steps 3] [--config c3|c5]      (c5: CIFAR d = 128, train-mode Hutchinson + CG; batch 32)
"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3", choices=["c3", "c5"]); ap.add_argument("--batch", type=int, default=None); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
args = ap.parse_args()
import bench
from cmf_amd import engine as E
from cmf_amd.optim import FlatOptimizer
dev = torch.device("cuda", 0)
args.batch = args.batch or (64 if args.config == "c3" else 32)
wl = bench.Workload(args.config, args.batch, 0, dev)
density, x = wl.inner, wl.x                                  # the noise is part of the synthetic input (as in bench.py)
density.train()
opt = FlatOptimizer(density.parameters(), opt="adam", lr=1e-4)
kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=wl.off, likelihood_wt=1., metric_wt=1.)

def step():
    opt.zero_grad()
    loss = -density.elbo(x.clone(), **kw)["elbo"].mean()
    loss.backward()
    opt.step()
    return loss

for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
_timing = E.timing(lambda name: True)
_timer = _timing.__enter__()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
by = _timer.by_name()
_timing.__exit__(None, None, None)
print(f"train step B={args.batch}: {1e3*dt:.1f} ms  ({args.batch/dt:.1f} samples/s, {1/dt:.3f} steps/s)  loss {float(loss.detach()):.1f}  "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
rows = [kv for kv in sorted(by.items(), key=lambda kv: -kv[1][1]) if kv[1][1] / args.steps >= 0.2] if by else []
print(f"  timed kernels in all: {sum(v[1] for v in by.values()) / args.steps:.1f} ms/step in {sum(v[0] for v in by.values()) // args.steps} launches")
for name, (n, ms, fl, _) in rows:
    print(f"  {name:40s} {n // args.steps:5d} launches/step  {ms/args.steps:9.2f} ms/step  {fl/max(ms,1e-9)/1e9:7.1f} TFLOP/s")
