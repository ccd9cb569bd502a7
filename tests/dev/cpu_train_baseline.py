#!/usr/bin/env python3
"""CPU reference point for the TRAINING step of the C3 model: loss.backward() through the float32 oracle (torch autograd, all
host threads).  Two flavours of the Jacobian assembly: "batched" (all d tangent columns at once: generous to the CPU) and
"ref_equivalent" (one decode per column with the primal recomputed: what the reference executes, non_square.py:298-311).

  python tests/dev/cpu_train_baseline.py [--batch 2] [--flavour batched]
"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=2); ap.add_argument("--flavour", default="batched")
args = ap.parse_args()
import cmf_amd
from cmf_amd.recipe import fill_state_dict
from oracle import cmf_oracle as O
torch.set_num_threads(os.cpu_count())
cfg = cmf_amd.get_config("mnist", latent_dimension=64, g_hidden_channels=[64] * 8, log_jacobian_method="cholesky")
schema, shape = cmf_amd.get_schema(cfg), cmf_amd.DATA_SHAPES["mnist"]
x = torch.randint(0, 256, (args.batch, *shape), generator=torch.Generator().manual_seed(1)).float()
dens = cmf_amd.get_density(schema, x)
sd = fill_state_dict(dens.state_dict(), seed=0)
named = dict(dens.named_parameters())
sd = {k: (v.clone().requires_grad_(True) if k in named else v) for k, v in sd.items()}
ops = O.compile_schema(schema, shape)
t0 = time.perf_counter()
elbo = O.elbo(sd, ops, x, add_offdiagonal_metric_reg=True, noise=torch.rand(x.shape), flavour=args.flavour)["elbo"]
t1 = time.perf_counter()
(-elbo.mean()).backward()
t2 = time.perf_counter()
print(f"CPU ({os.cpu_count()} threads, {args.flavour}) B={args.batch}: forward {t1 - t0:.1f} s, backward {t2 - t1:.1f} s -> "
      f"{args.batch / (t2 - t0):.3f} samples/s")
