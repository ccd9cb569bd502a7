#!/usr/bin/env python3
"""C5 training step (CIFAR d=128, 32 samples, train-mode Hutchinson S=4 + CG): low-rank backward with 16 / 32 column slots against
the d-column backward.   python tools/exp_c5_train.py [--B 32] [--steps 3]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from cmf_amd.optim import FlatOptimizer
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=32); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--modes", default="16,32,full")
a = ap.parse_args()
dev = torch.device("cuda", 0)
for mode in a.modes.split(","):
    wl = bench.Workload("c5", a.B, 0, dev)
    head = [m for m in wl.density.modules() if type(m).__name__ == "NonSquareHeadDensity"][0]
    if mode == "full":
        head.hutch_lowrank = False
    else:
        head.HUTCH_LOWRANK_NC = int(mode)
    wl.density.train()
    opt = FlatOptimizer(wl.density.parameters(), opt="adam", lr=1e-4)
    def step():
        opt.zero_grad()
        loss = -wl.inner.elbo(wl.x, **wl.kw)["elbo"].mean()
        loss.backward(); opt.step()
        return loss
    torch.cuda.reset_peak_memory_stats()
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps): l = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(f"C5 train B={a.B} mode={mode:5s}: {ms:7.1f} ms/step  {a.B / ms * 1e3:7.1f} samples/s  peak {torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB  loss {float(l):.4f}", flush=True)
    del opt, wl, head
    torch.cuda.empty_cache()
