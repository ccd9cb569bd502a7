#!/usr/bin/env python3
"""Run the other BASELINE configs at full size on the GPU: finite outputs, timing, and (small batches) oracle check."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import cmf_amd
from cmf_amd.recipe import fill_state_dict
from oracle import cmf_oracle as O

def run(name, dataset, over, B, check_B, off=True, reps=5):
    cfg = cmf_amd.get_config(dataset, **over)
    schema = cmf_amd.get_schema(cfg)
    shape = cmf_amd.DATA_SHAPES[dataset]
    gen = torch.Generator().manual_seed(1234)
    if len(shape) == 3:
        x = torch.randint(0, 256, (B, *shape), generator=gen).float() + torch.rand(B, *shape, generator=gen)
    else:
        x = torch.randn(B, *shape, generator=gen)
        if dataset == "sphere": x = x / x.norm(dim=1, keepdim=True)
    dens = cmf_amd.get_density(schema, x)
    sd = fill_state_dict(dens.state_dict(), 0); dens.load_state_dict(sd); dens = dens.cuda().eval()
    inner = dens.module.density if schema[0]["type"] == "dequantization" else dens
    xg = x.cuda()
    with torch.no_grad():
        out = inner.elbo(xg, add_reconstruction=True, add_offdiagonal_metric_reg=off)["elbo"]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): inner.elbo(xg, add_reconstruction=True, add_offdiagonal_metric_reg=off)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        from cmf_amd.graphs import ElboGraph
        eg = ElboGraph(inner, xg, add_reconstruction=True, add_offdiagonal_metric_reg=off)
        eg(xg); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps * 4): eg(xg)
        torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / (reps * 4)
        gerr = float((eg(xg)["elbo"] - out).abs().max() / out.abs().max())
        rel = None
        if check_B:
            ops = O.compile_schema(schema, shape)
            want = O.elbo(sd, ops, x[:check_B], add_offdiagonal_metric_reg=off, noise=torch.zeros_like(x[:check_B]))["elbo"]
            rel = float((out[:check_B].cpu() - want).abs().max() / want.abs().max())
    print(json.dumps({"config": name, "B": B, "ms": 1e3 * dt, "evals_per_s": B / dt, "graph_ms": 1e3 * dtg, "graph_evals_per_s": B / dtg, "graph_vs_eager": gerr, "finite": bool(torch.isfinite(out).all()),
                      "rel_err_vs_oracle": rel, "mem_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)

which = sys.argv[1:] or ["C1", "C2a", "C2b", "C5"]
if "C1" in which: run("C1 sphere D=3 d=3", "sphere", {"latent_dimension": 3}, 1024, 64)
if "C2a" in which: run("C2a power D=6 d=2", "power", {}, 4096, 64)
if "C2b" in which: run("C2b hepmass D=21 d=10", "hepmass", {}, 4096, 64)
if "C5" in which: run("C5 cifar10 D=3072 d=128 (exact/eval path), per-GPU shard B=32", "cifar10", {"latent_dimension": 128, "hutchinson_samples": 4}, 32, 1, off=False, reps=2)
if "C5full" in which: run("C5 cifar10 D=3072 d=128 B=256 on one GPU", "cifar10", {"latent_dimension": 128, "hutchinson_samples": 4}, 256, 0, off=False, reps=1)
