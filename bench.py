#!/usr/bin/env python3
"""Headline benchmark: log-density evals/sec (J^T J-Cholesky path), MNIST D=784 d=64 bs=512.

One step = one ``density.elbo(x, add_reconstruction=True, add_offdiagonal_metric_reg=True,
likelihood_wt=1, metric_wt=1)`` in eval mode under no_grad on a synthetic batch resident in HBM
(SURVEY.md section 8d), followed -- when N > 1 -- by the RCCL all-reduce of (sum elbo, count).
One process per GPU; every rank evaluates its own B samples (weak scaling: samples are independent,
there is no data-path collective).  ``--strong`` shards a fixed global batch of 512 instead
(BASELINE.json configs[3]).

Prints ONE JSON line on rank 0, with
  roofline      dominant kernel (3x3 64->64 tangent convolution on fp32 MFMA) timed live with HIP
                events on the launch stream over the timed steps
  cpu_baseline  the CPU oracle's reference-equivalent flavour (column loop with primal recompute)
                timed on this host's cores on a bounded sample (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0                  # HBM3E spec (6.3 TB/s measured achievable)


def make_model(device, d=64, hidden=(64,) * 8, dataset="mnist", seed=0):
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    cfg = cmf_amd.get_config(dataset, latent_dimension=d, g_hidden_channels=list(hidden), log_jacobian_method="cholesky")
    schema = cmf_amd.get_schema(cfg)
    shape = cmf_amd.DATA_SHAPES[dataset]
    density = cmf_amd.get_density(schema, torch.zeros(1, *shape))
    sd = fill_state_dict(density.state_dict(), seed=seed)        # identical weights + permutation on every rank
    density.load_state_dict(sd)
    return cfg, schema, shape, sd, density.to(device).eval()


def synth_batch(shape, B, rank, device):
    gen = torch.Generator().manual_seed(1234 + rank)
    x = torch.randint(0, 256, (B, *shape), generator=gen).float()
    return x.to(device)


def cpu_baseline(schema, shape, sd, B_cpu):
    """Reference-equivalent CPU restatement (oracle, kind 'port') on a bounded sample."""
    from oracle import cmf_oracle as O
    ops = O.compile_schema(schema, shape)
    gen = torch.Generator().manual_seed(99)
    x = torch.randint(0, 256, (B_cpu, *shape), generator=gen).float()
    noise = torch.rand(x.shape, generator=gen)
    # the GPU box exposes the whole host in os.cpu_count() but grants a 16-CPU share per GPU:
    # oversubscribed OpenMP teams crawl, so size the pool to the affinity mask capped at that share
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("CMF_CPU_THREADS", 16))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle ref-equivalent, B={B_cpu}, {cores} threads ...", file=sys.stderr, flush=True)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.elbo(sd, ops, x, add_offdiagonal_metric_reg=True, noise=noise, flavour="ref_equivalent")
        dt = time.perf_counter() - t0
    return {"value": B_cpu / dt, "unit": "evals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B={B_cpu} full model (d=64, 10 ResNet couplers), one elbo call, {dt:.1f} s, "
                      f"oracle.jtj_ref_equivalent (column loop + primal recompute = what the reference executes)"}


def pmc_traffic(precision):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this same
    command (profiles/*_pmc_*.json; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction), or None."""
    path = os.path.join(ROOT, "profiles", f"pmc_conv_tangent_{precision}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("traffic_bytes_per_launch")


def train_bench(args, density, x, B, rank, world, device):
    """Secondary metric: training samples / s (one process per GPU, flat gradient all-reduce over RCCL, fused Adam)."""
    import torch.distributed as dist
    from cmf_amd.optim import FlatOptimizer
    density.train()
    opt = FlatOptimizer(density.parameters(), opt="adam", lr=1e-4)
    kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=True, likelihood_wt=1., metric_wt=1.)

    def step():
        opt.zero_grad()
        loss = -density.elbo(x, **kw)["elbo"].mean()
        loss.backward()
        opt.allreduce_flat()
        opt.step()
        return loss.detach()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    if rank == 0:
        print(json.dumps({
            "metric": "training samples/sec (forward + backward + Adam, JtJ-cholesky objective), MNIST D=784 d=64", "value": B * world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (3x3 tangent convs, their transposes and weight gradients as bf16x3 split MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "C3 / C4 model, one optimiser step per batch, g_ij off-diagonal objective", "per_gpu_batch": B,
                       "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(loss),
                       "peak_memory_gib": torch.cuda.max_memory_allocated(device) / 2 ** 30}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="samples per GPU (weak) or global batch (--strong)")
    ap.add_argument("--strong", action="store_true", help="shard a fixed global batch over the ranks")
    ap.add_argument("--cpu-batch", type=int, default=32, help="CPU baseline sample size (0 disables); 32 = ~12 s on 16 cores")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--precision", choices=["bf16x3", "f32"], default="bf16x3",
                    help="arithmetic of the 3x3 tangent convolutions (both are fp32-grade; see DESIGN.md 4.5)")
    ap.add_argument("--primal-precision", choices=["f32", "bf16x3"], default="f32",
                    help="arithmetic of the PRIMAL hidden convs (relu masks come from these activations): f32 = exact fp32 "
                         "products (default, parity-grade), bf16x3 = split precision, ~9 %% faster (DESIGN.md 4.2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 "
                                                       "on a one-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (never a measurement)")
    ap.add_argument("--train", action="store_true",
                    help="SECONDARY metric (SURVEY 8d): time training steps instead -- forward + loss.backward() on the HIP kernels + "
                         "data-parallel gradient all-reduce + fused Adam; use --batch 64 (the reference's per-GPU shard)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    device = torch.device("cuda", 0 if args.share_gpu else local)
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from cmf_amd import engine as E
    from cmf_amd.distributed import allreduce_mean_elbo
    E.TANGENT_PRECISION = args.precision
    E.PRIMAL_PRECISION = args.primal_precision
    cfg, schema, shape, sd, density = make_model(device)
    inner = density.module.density                       # feed dequantised data ourselves: noise is part of the synthetic input
    B = args.batch // world if args.strong else args.batch
    x = synth_batch(shape, B, rank, device)
    x = x + torch.rand(x.shape, generator=torch.Generator().manual_seed(4321 + rank)).to(device)

    def step():
        out = inner.elbo(x, add_reconstruction=True, add_offdiagonal_metric_reg=True, likelihood_wt=1., metric_wt=1.)
        return allreduce_mean_elbo(out["elbo"])          # (sum, count) all-reduce; plain mean on one rank

    if args.train:
        return train_bench(args, inner, x, B, rank, world, device)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        if not args.no_kernel_timer:
            E.TIMER = E.KernelTimer(lambda name: name == "conv_tangent_t9_ci64_co64")
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        dt = time.perf_counter() - t0
    ksum = E.TIMER.summary() if E.TIMER is not None else None
    E.TIMER = None

    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        total = B * world * args.steps
        line = {
            "metric": "log-density evals/sec (JtJ-cholesky path), MNIST D=784 d=64 bs=512",
            "value": total / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f32 (3x3 tangent convs as bf16x3 split MFMA, fp32 accumulate)",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, not a measurement)" if args.share_gpu else ""),
            "config": {"workload": "C3: MNIST-shaped (1,28,28) uint8-range + U[0,1) noise, non-square flow d=64, "
                                   "cholesky J^T J log-det + g_ij off-diagonal L1 + reconstruction, eval/no_grad",
                       "per_gpu_batch": B, "global_batch": B * world, "D": 784, "latent_dimension": 64,
                       "parallelism": f"dp{world}", "loss_mean": float(loss)},
        }
        if ksum and ksum["launches"]:
            sec = ksum["total_ms"] * 1e-3
            tf, gbs = ksum["flops"] / sec / 1e12, ksum["bytes"] / sec / 1e9
            common = {"launches": ksum["launches"], "avg_ms": ksum["total_ms"] / ksum["launches"],
                      "share_of_step": ksum["total_ms"] / (1e3 * dt), "traffic": pmc_traffic(E.TANGENT_PRECISION),
                      "algorithmic_tflops": tf, "algorithmic_gbs": gbs}
            if E.TANGENT_PRECISION == "bf16x3":
                # Split precision: every fp32-grade product is THREE bf16 MFMA products (hi*hi + hi*lo + lo*hi), so the
                # matrix work this algorithm needs is 3x the algorithmic fp32 flops; `achieved` counts exactly that
                # (nothing else: the K packing has no zero-weight padding) against the dense bf16 peak.
                # Measured with in-kernel stamps the kernel is bound by SIMD issue (MFMA + the loader waves' VALU), not
                # by HBM: `hbm_view` carries the memory side.
                line["roofline"] = {"bound": "mfma", "achieved": 3.0 * tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": 3.0 * tf / BF16_MFMA_PEAK_TFLOPS,
                                    "kernel": "conv_tangent_bf16x3_kernel<4,7,3> (<4,7,1> for the first hidden conv of each coupler; 3x3, "
                                              "64->64 channels, all d Jacobian columns; split-precision bf16 MFMA, fp32 accumulate)",
                                    "fp32_equivalent_tflops": tf, "bf16_products_per_fp32_product": 3,
                                    "hbm_view": {"algorithmic_gbs": gbs, "peak_gbs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS},
                                    **common}
            else:
                line["roofline"] = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                                    "kernel": "conv_tangent_kernel<9,4,7> (3x3, 64->64 channels, all d Jacobian columns; fp32 MFMA)",
                                    **common}
        if world == 1 and args.cpu_batch > 0:
            line["cpu_baseline"] = cpu_baseline(schema, shape, {k: v.cpu() for k, v in sd.items()}, args.cpu_batch)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
