#!/usr/bin/env python3
"""Headline benchmark: log-density evals/sec (J^T J-Cholesky path), MNIST D=784 d=64 bs=512.

One step = one ``density.elbo(x, add_reconstruction=True, add_offdiagonal_metric_reg=True,
likelihood_wt=1, metric_wt=1)`` in eval mode under no_grad on a synthetic batch resident in HBM
(SURVEY.md section 8d), followed -- when N > 1 -- by the RCCL all-reduce of (sum elbo, count).
One process per GPU; every rank evaluates its own B samples (weak scaling: samples are independent,
there is no data-path collective).  ``--strong`` shards a fixed global batch instead
(BASELINE.json configs[3]).

Launching.  ``python bench.py --gpus N`` from a bare shell starts its own N rank processes (the
parent never touches the GPU: it only parses the arguments, spawns fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and returns non-zero if any
child fails) -- the one-command, N-devices form of the reference's ``nn.DataParallel`` wrapper
(wrapper.py:52-68).  Under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``
the environment already carries the rank, and this process IS a rank.

Prints ONE JSON line on rank 0, with
  roofline      dominant kernel timed live with HIP events on the launch stream over the timed steps
  cpu_baseline  the CPU oracle's reference-equivalent flavour (column loop with primal recompute)
                timed on this host's cores on a bounded sample (rank 0, N = 1 only)
  f32_exact     (default workload, N = 1) the same step with the 3x3 tangent convs on exact-fp32 MFMA
  ranks_seen    world size as seen by the process group after the timing all-reduce

``--config {c1,c2a,c2b,c5}`` runs the other BASELINE configurations through the same contract.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0         # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0                  # HBM3E spec (6.3 TB/s measured achievable)

#: BASELINE.json configs -> (dataset, config overrides, default per-GPU batch, off-diagonal metric term, label)
CONFIGS = {
    "c1": ("sphere", {"latent_dimension": 3}, 1024, True, "C1: sphere D=3 d=3, 5 ACL MLP[10,10] + affine prior"),
    "c2a": ("power", {}, 4096, True, "C2a: power-shaped tabular D=6 d=2, 10 ACL MLP[128]x4 + realnvp prior"),
    "c2b": ("hepmass", {}, 4096, True, "C2b: hepmass-shaped tabular D=21 d=10, 10 ACL MLP[128]x4 + realnvp prior"),
    "c3": ("mnist", {"latent_dimension": 64, "log_jacobian_method": "cholesky"}, 512, True,
           "C3: MNIST-shaped (1,28,28) uint8-range + U[0,1) noise, non-square flow d=64, cholesky J^T J log-det + "
           "g_ij off-diagonal L1 + reconstruction, eval/no_grad"),
    "c5": ("cifar10", {"latent_dimension": 128, "hutchinson_samples": 4}, 32, False,
           "C5: CIFAR-shaped (3,32,32) D=3072 d=128, 32 samples per GPU (256 over 8)"),
}
#: SURVEY 8d, "Algorithmic work per eval (dense; the figure roofline.achieved uses)": W_alg = (2 + d) F_net + F_prior + 2 D d^2 + d^3/3, FLOP / sample
W_ALG_FLOP = {"c1": 7.3e3, "c2a": 4.0e6, "c2b": 12.8e6, "c3": 290.42e9, "c5": 749.67e9}
METRICS = {
    "c1": "log-density evals/sec (JtJ-cholesky path), sphere D=3 d=3 bs=1024",
    "c2a": "log-density evals/sec (JtJ-cholesky path), tabular D=6 d=2 bs=4096",
    "c2b": "log-density evals/sec (JtJ-cholesky path), tabular D=21 d=10 bs=4096",
    "c3": "log-density evals/sec (JtJ-cholesky path), MNIST D=784 d=64 bs=512",
    "c5": "log-density evals/sec, CIFAR-10 D=3072 d=128 bs=32 per GPU",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; 20 for the launch-bound c1 / c2)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE.json configuration (default: the headline C3)")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (weak) or global batch (--strong); default per config")
    ap.add_argument("--strong", action="store_true", help="shard a fixed global batch over the ranks")
    ap.add_argument("--cpu-batch", type=int, default=None, help="CPU baseline sample size (0 disables); C3: 32 = ~12 s on 16 cores")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-f32-exact", action="store_true", help="skip the exact-fp32 leg of the default C3 run")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph (default for the launch-bound "
                                                          "c1 / c2a / c2b); the roofline leg then comes from separate eager steps")
    ap.add_argument("--no-graph", action="store_true", help="c1 / c2a / c2b: time eager launches instead of graph replays")
    ap.add_argument("--hutchinson", action="store_true", help="c5: train-mode stochastic log-det (Hutchinson S=4 + CG) instead of the "
                                                               "exact eval path")
    ap.add_argument("--precision", choices=["bf16x3", "f32"], default="bf16x3",
                    help="arithmetic of the 3x3 tangent convolutions (both are fp32-grade; see DESIGN.md 4.1b)")
    ap.add_argument("--primal-precision", choices=["f16x3", "f32", "bf16x3"], default="f16x3",
                    help="arithmetic of the PRIMAL hidden convs (relu masks come from these activations): f16x3 = fp16 split with "
                         "exact power-of-two scales (default: fp32-grade per product, bf16 MFMA rate), f32 = exact fp32 products, "
                         "bf16x3 = bf16 split (experiment: ~15x more relu-mask flips; DESIGN.md 4.2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1 "
                                                       "on a one-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0 (never a measurement)")
    ap.add_argument("--no-legs", action="store_true", help="default C3 run only: skip the secondary measurements (N = 1: c5, c2b, "
                                                            "c5_train, train; N > 1: strong_c3, c5)")
    ap.add_argument("--leg-steps", type=int, default=10, help="timed steps of each secondary measurement")
    ap.add_argument("--train-batch", type=int, default=64, help="samples per GPU of the C3 training leg (64 = the reference's per-GPU "
                                                                 "shard, BASELINE configs[3]; smaller only to rehearse many ranks on one GPU)")
    ap.add_argument("--force-group", action="store_true",
                    help="N = 1: build a ONE-rank process group anyway and route the timing through its collectives (barrier, "
                         "(sum, count) all-reduce, MAX of the times) -- RCCL itself on a one-GPU box (tests/test_gpu_round4.py)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="--gpus N from a bare shell: seconds before the parent "
                                                                          "stops every rank and returns 124")
    ap.add_argument("--train", action="store_true",
                    help="SECONDARY metric (SURVEY 8d): time training steps instead -- forward + loss.backward() on the HIP kernels + "
                         "data-parallel gradient all-reduce + fused Adam; use --batch 64 (the reference's per-GPU shard)")
    args = ap.parse_args(argv)
    if args.config in ("c1", "c2a", "c2b") and not args.no_graph and not args.train:
        # the launch-bound configurations replay a captured HIP graph.  C3 / C5 stay eager by default: measured in round 3, a
        # replayed C3 step takes 432.0 ms against 434.7 ms eager (C5 shard 75.3 against 75.6) -- the GPU is busy either way -- and
        # eager launches keep the dominant kernel's HIP events INSIDE the timed region (--graph replays them too)
        args.graph = True
    if args.steps is None:
        args.steps = 20 if args.config in ("c1", "c2a", "c2b") else 3
    if args.batch is None:
        args.batch = CONFIGS[args.config][2]
    if args.cpu_batch is None:
        args.cpu_batch = {"c1": 1024, "c2a": 4096, "c2b": 4096, "c3": 32, "c5": 2}[args.config]
    return args


# ----------------------------------------------------------------------------------------------------------------------
# parent: spawn one fresh process per GPU (never initialises the GPU itself, never exec()s)
# ----------------------------------------------------------------------------------------------------------------------


def spawn_ranks(args, argv, script=None):
    """Start one fresh process per GPU, relay rank 0's JSON line, return the exit code.  Every child is watched: the first one
    that exits non-zero (e.g. RCCL could not initialise: run_rank prints the reason and exits 3) stops the others at once --
    they would otherwise sit in the rendezvous or a collective until the backend's own timeout -- and an overall deadline
    (--launch-timeout) bounds the whole run.  Nothing is ever re-executed in a process that has touched the GPU."""
    import tempfile
    import threading
    attempt = getattr(args, "_launch_attempt", 0)
    t_start = time.monotonic()
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]                      # only a fallback now (MASTER_PORT for code that insists on env://)
    # Rendezvous through a FILE store (round 5): the ranks of this launcher meet in a fresh file under a private temporary directory
    # (init_group: init_method="file://..."), so there is no TCP port to probe, release and race for.  Under
    # torch.distributed.run the environment carries MASTER_ADDR / MASTER_PORT instead and the ranks use env:// as before.
    rdv_dir = tempfile.mkdtemp(prefix="cmf_bench_rdv_")
    rdv = os.path.join(rdv_dir, f"store_{attempt}")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CMF_RENDEZVOUS_FILE=rdv)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "2")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(args.launch_timeout)
    timed_out = False
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
            break
        if time.monotonic() > deadline:
            timed_out = True
            break
        time.sleep(0.1)
    for p in procs:                                    # stop whoever is still running (only after a failure or the deadline)
        if p.poll() is None:
            p.terminate()
    for p in procs:
        try:
            p.wait(timeout=15)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    reader.join(timeout=5)
    import shutil
    shutil.rmtree(rdv_dir, ignore_errors=True)
    codes = [p.returncode for p in procs]
    lines = [l for l in ("".join(o or "" for o in out0)).splitlines() if l.strip()]
    for l in lines:                                    # the JSON line to stdout; anything a backend chattered (gloo does) to stderr
        print(l, flush=True, file=sys.stdout if l.lstrip().startswith("{") else sys.stderr)
    if timed_out:
        print(f"[bench] ranks still running after --launch-timeout {args.launch_timeout:.0f} s: stopped (codes {codes})", file=sys.stderr)
        return 124
    if 3 in codes and attempt == 0 and time.monotonic() - t_start < 90 and not any(l.lstrip().startswith("{") for l in lines):
        # exit code 3 = init_group could not build the process group (run_rank): within the first seconds that is the rendezvous.
        # Fresh children on a fresh store file, once.
        print(f"[bench] process group failed within {time.monotonic() - t_start:.0f} s (codes {codes}): retrying once with a fresh rendezvous",
              file=sys.stderr)
        args._launch_attempt = 1
        return spawn_ranks(args, argv, script)
    if any(codes):
        print(f"[bench] rank exit codes {codes} (a failing rank stops the others; its own message is above)", file=sys.stderr)
        return next(c for c in codes if c and c > 0) if any(c and c > 0 for c in codes) else 1
    if not any(l.lstrip().startswith("{") for l in lines):
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------


def make_model(device, d=64, hidden=(64,) * 8, dataset="mnist", seed=0, overrides=None):
    import torch
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    if overrides is None:
        overrides = dict(latent_dimension=d, g_hidden_channels=list(hidden), log_jacobian_method="cholesky")
    cfg = cmf_amd.get_config(dataset, **overrides)
    schema = cmf_amd.get_schema(cfg)
    shape = cmf_amd.DATA_SHAPES[dataset]
    density = cmf_amd.get_density(schema, torch.zeros(1, *shape))
    sd = fill_state_dict(density.state_dict(), seed=seed)        # identical weights + permutation on every rank
    density.load_state_dict(sd)
    return cfg, schema, shape, sd, density.to(device).eval()


def synth_batch(dataset, shape, B, rank, device):
    """SURVEY 8d's synthetic inputs; the dequantisation noise is part of the input (same generator)."""
    import torch
    gen = torch.Generator().manual_seed(1234 + rank)
    if len(shape) == 3:
        x = torch.randint(0, 256, (B, *shape), generator=gen).float()
        x = x + torch.rand(x.shape, generator=torch.Generator().manual_seed(4321 + rank))
    else:
        x = torch.randn(B, *shape, generator=gen)
        if dataset == "sphere":
            x = x / x.norm(dim=1, keepdim=True)
    return x.to(device)


def cpu_baseline(schema, shape, sd, B_cpu, dataset, off, label):
    """Reference-equivalent CPU restatement (oracle, kind 'port') on a bounded sample."""
    import torch
    from oracle import cmf_oracle as O
    ops = O.compile_schema(schema, shape)
    gen = torch.Generator().manual_seed(99)
    if len(shape) == 3:
        x = torch.randint(0, 256, (B_cpu, *shape), generator=gen).float()
        noise = torch.rand(x.shape, generator=gen)
    else:
        x = torch.randn(B_cpu, *shape, generator=gen)
        if dataset == "sphere":
            x = x / x.norm(dim=1, keepdim=True)
        noise = None
    # the GPU box exposes the whole host in os.cpu_count() but grants a 16-CPU share per GPU:
    # oversubscribed OpenMP teams crawl, so size the pool to the affinity mask capped at that share
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("CMF_CPU_THREADS", 16))))
    torch.set_num_threads(cores)
    # SURVEY 8d / BASELINE.md section 3: one untimed warm-up call (thread pool, oneDNN primitive caches, page faults), then the cost
    # at B_cpu / 4, B_cpu / 2, B_cpu (C3: 8, 16, 32 -- linear in B); `value` is the largest sample's rate
    sizes = [B_cpu] if B_cpu < 8 else [B_cpu // 4, B_cpu // 2, B_cpu]
    warm = 0 if B_cpu < 8 else max(1, sizes[0] // 2)
    print(f"[bench] cpu_baseline: oracle ref-equivalent, warm-up B={warm}, timed B={sizes}, {cores} threads ...", file=sys.stderr, flush=True)
    run = lambda n: O.elbo(sd, ops, x[:n], add_offdiagonal_metric_reg=off, noise=None if noise is None else noise[:n], flavour="ref_equivalent")
    per_b = {}
    with torch.no_grad():
        if warm:
            run(warm)
        for n in sizes:
            t0 = time.perf_counter()
            run(n)
            per_b[n] = time.perf_counter() - t0
    dt = per_b[B_cpu]
    return {"value": B_cpu / dt, "unit": "evals/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
            "evals_per_s_by_batch": {str(n): n / t for n, t in per_b.items()}, "warmup_batch": warm,
            "sample": f"B={B_cpu}, {label}, one elbo call after a warm-up call at B={warm}, {dt:.1f} s ({sum(per_b.values()):.0f} s for "
                      f"B = {sizes}), oracle.jtj_ref_equivalent (column loop + primal recompute = what the reference executes)"}


def cpu_model():
    """'model name' of /proc/cpuinfo ("state core count and CPU model")."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


#: libcmf_amd symbol -> stage of the per-stage breakdown (calls inside FlowProgram.encode all count as "encode")
STAGE_OF = {"cmf_conv_tangent_bf16x3": "tangent_hidden", "cmf_conv_tangent": "tangent_first_last",
            "cmf_conv_tangent_f16x3": "primal", "cmf_conv_tangent:primal": "primal", "cmf_conv_tangent_bf16x3:primal": "primal",
            "cmf_conv_primal": "primal", "cmf_primal_regroup": "primal", "cmf_absmax": "primal",
            "cmf_acl_tangent": "acl", "cmf_acl_primal": "acl", "cmf_gram_cholesky": "gram_cholesky", "cmf_cholesky_retry": "gram_cholesky",
            "cmf_mlp_coupler": "mlp_coupler"}
STAGES = ("encode", "primal", "tangent_hidden", "tangent_first_last", "acl", "gram_cholesky", "other", "host_gap")


def stages_leg(wl, ms_per_step, steps=2):
    """SURVEY 8d "per-stage times (encode, decode + tangents, Gram, Cholesky, reductions)": ``steps`` eager evaluations AFTER the timed
    region with EVERY libcmf_amd launch bracketed by HIP events (cmf_amd._lib.trace), folded into stages by symbol and by the
    FlowProgram phase the call was made in.  ms / step; ``host_gap`` = the timed region's ms_per_step minus the kernels' sum (launch
    gaps, torch's own fill / copy kernels, the event overhead's mirror image), so the stages sum to ms_per_step."""
    import torch
    from cmf_amd import _lib
    with torch.no_grad():
        wl.inner.elbo(wl.x, **wl.kw)
        torch.cuda.synchronize()
        with _lib.trace() as rec:
            for _ in range(steps):
                wl.inner.elbo(wl.x, **wl.kw)
            torch.cuda.synchronize()
    out = {k: 0.0 for k in STAGES}
    launches = {k: 0 for k in STAGES}
    for name, ph, e0, e1 in rec:
        st = "encode" if ph in ("encode", "encode_nested") else STAGE_OF.get(name, "other")
        if st == "mlp_coupler":
            st = "tangent_hidden"                          # flat models: the fused coupler kernel IS the tangent pass
        out[st] += e0.elapsed_time(e1) / steps
        launches[st] += 1
    kernels = sum(out.values())
    out["host_gap"] = ms_per_step - kernels
    return {"unit": "ms/step", **{k: round(v, 4) for k, v in out.items()}, "kernels_sum": round(kernels, 4),
            "launches_per_step": {k: v // steps for k, v in launches.items() if v},
            "note": f"HIP events around every libcmf_amd launch of {steps} eager steps after the timed region; primal = the coupler "
                    "networks' primal pass of the decode side (first conv, regroup, hidden convs, 1x1 + ScaledTanh), encode = the "
                    "whole encode pass; host_gap = ms_per_step - kernels_sum"}


def pmc_traffic(precision, B):
    """HBM-side bytes per launch of the dominant kernel: the COMMITTED rocprofv3 --pmc passes of the default command
    (profiles/pmc_conv_tangent_*.json, B = 512; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction), scaled
    linearly to this run's per-GPU batch (the kernel's traffic is per sample).  Not re-measured by this run."""
    path = os.path.join(ROOT, "profiles", f"pmc_conv_tangent_{precision}.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        j = json.load(f)
    t = j.get("traffic_bytes_per_launch")
    if t is None:
        return None, None
    # the passes were taken on ONE version of the kernel: say whether it is the source this run was built from
    measured, current = j.get("kernel_source_sha16"), kernel_source_sha16(precision)
    same = {True: "same kernel source as this build", False: "STALE: the kernel source has changed since (re-run tools/refresh_profiles.sh)",
            None: "kernel source version not recorded"}[None if measured is None else measured == current]
    return t * B / float(j.get("batch", 512)), (f"committed PMC passes (profiles/{os.path.basename(path)}, B={j.get('batch', 512)}, kernel "
                                                f"source sha16 {measured}: {same}) scaled to per_gpu_batch={B}; not collected by this run")


def kernel_source_sha16(precision):
    """First 16 hex digits of the SHA-256 of the dominant kernel's source file (stamped into the PMC summaries by
    tools/pmc_kernel.py --source)."""
    import hashlib
    src = os.path.join(ROOT, "cmf_amd", "csrc", "conv_tangent_bf16x3.hip" if precision == "bf16x3" else "conv_tangent.hip")
    with open(src, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def grad_reduce_leg(n, rank, world, device, iters=20, warmup=3):
    """N > 1 only (gradient reduction shaped for the point-to-point xGMI mesh): the flat gradient bucket of the C3 model
    (``n`` floats + the sample-count word) summed over the ranks as ONE all-reduce and as reduce-scatter + all-gather
    (cmf_amd.distributed.sum_flat), ``iters`` calls each between barrier + synchronize brackets, max over ranks.  The training
    legs of the line use the faster shape."""
    import torch
    import torch.distributed as dist
    from cmf_amd.distributed import sum_flat, rs_ag_scratch, REDUCE_SHAPES
    bucket = torch.zeros(n + 4, dtype=torch.float32, device=device)
    scratch = torch.empty(rs_ag_scratch(n + 4), dtype=torch.float32, device=device)
    us = {}
    for shape in REDUCE_SHAPES:
        for _ in range(warmup):
            sum_flat(bucket, shape, scratch)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            sum_flat(bucket, shape, scratch)
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        us[shape] = 1e6 * float(dt.item()) / iters
    best = min(REDUCE_SHAPES, key=lambda k: us[k])             # the same on every rank: the times are the max over ranks
    return {"unit": "us per reduction", "bytes": 4 * (n + 4), "iters": iters, **{k: us[k] for k in REDUCE_SHAPES}, "used": best,
            "bus_gbs": {k: 2.0 * (world - 1) / world * 4 * (n + 4) / (us[k] * 1e-6) / 1e9 for k in REDUCE_SHAPES},
            "note": "bus_gbs = 2 (N-1)/N x bytes / time, the ring-equivalent bus bandwidth; one collective after the backward pass, "
                    "not overlapped with it (a 24 MB bucket against a 185 ms step)"}


def train_leg(steps, warmup, config, density, x, B, rank, world, device, off=True, reduce_shape="all_reduce"):
    """Secondary metric: training samples / s (one process per GPU, flat gradient all-reduce over RCCL, fused Adam).  C3: the
    Cholesky objective with the off-diagonal metric term; C5: train mode selects the Hutchinson + CG surrogate
    (non_square.py:131-138), whose backward runs through the 2 S probe directions.  Returns the result object (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from cmf_amd.optim import FlatOptimizer
    density.train()
    opt = FlatOptimizer(density.parameters(), opt="adam", lr=1e-4, reduce_shape=reduce_shape)
    kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=off, likelihood_wt=1., metric_wt=1.)

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

    torch.cuda.reset_peak_memory_stats(device)
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    density.eval()
    peak = torch.cuda.max_memory_allocated(device) / 2 ** 30
    del opt
    if rank != 0:
        return None
    return {
        "metric": ("training samples/sec (forward + backward + Adam, JtJ-cholesky objective), MNIST D=784 d=64" if config == "c3" else
                   f"training samples/sec (forward + backward + Adam), {METRICS[config].split(', ', 1)[1]}"),
        "value": B * world * steps / dt,
        "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (3x3 tangent convs, their transposes and weight gradients as bf16x3 split MFMA, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": ("C3 / C4 model, one optimiser step per batch, g_ij off-diagonal objective" if config == "c3" else
                                CONFIGS[config][4] + ", one optimiser step per batch, train-mode objective (Hutchinson S=4 + CG)"),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(loss),
                   "peak_memory_gib": peak, "gradient_reduction": reduce_shape if world > 1 else None}}


def guarded(name, fn, world):
    """A SECONDARY measurement must not cost the headline: on one rank an exception inside it is printed to stderr and recorded as
    ``{"error": ...}`` in its place (the headline was timed before any leg ran).  With more ranks the legs contain collectives: a rank
    that skipped one would leave the others waiting in it, so there the exception ends the rank and the launcher stops the job."""
    if world > 1:
        return fn()
    try:
        return fn()
    except Exception as e:                                  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        print(f"[bench] secondary measurement {name!r} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        try:
            import torch
            torch.cuda.empty_cache()
        except Exception:                                   # noqa: BLE001
            pass
        return {"error": f"{type(e).__name__}: {e}"[:400]}


def dominant(rows):
    """Kernel family with the largest summed duration: (name, launches, total_ms, flops, bytes)."""
    name = max(rows, key=lambda k: rows[k][1])
    return (name, *rows[name])


def roofline_of(name, n, ms, flops, nbytes, step_ms_total, precision, B, pmc=True):
    """Roofline object for one kernel family from its HIP-event totals."""
    sec = ms * 1e-3
    tf, gbs = flops / sec / 1e12, nbytes / sec / 1e9
    common = {"launches": n, "avg_ms": ms / n, "share_of_step": ms / step_ms_total,
              "algorithmic_tflops": tf, "algorithmic_gbs": gbs, "timer_name": name}
    split = precision == "bf16x3" and name == "conv_tangent_t9_ci64_co64"
    if split:
        # Split precision: every fp32-grade product is THREE bf16 MFMA products (hi*hi + hi*lo + lo*hi), so the matrix work
        # this algorithm needs is 3x the algorithmic fp32 flops; `achieved` counts exactly that (nothing else: the K packing
        # has no zero-weight padding) against the dense bf16 peak.  `hbm_view` carries the memory side.
        traffic, note = pmc_traffic("bf16x3", B) if pmc else (None, "PMC passes were taken on the C3 command only")
        return {"bound": "mfma", "achieved": 3.0 * tf, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": 3.0 * tf / BF16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": note,
                "kernel": "conv_tangent_bf16x3_kernel (3x3, 64->64 channels, all d Jacobian columns; split-precision bf16 MFMA, "
                          "fp32 accumulate)",
                "fp32_equivalent_tflops": tf, "bf16_products_per_fp32_product": 3,
                # context, not the contract's peak: what a loop of nothing but this MFMA sustains on live random operands on every
                # SIMD (tools/ubench/mfmapower.py, profiles/r03_mfma_sustained.txt: the chip lowers its clock under the load)
                "live_data_mfma_ceiling": {"frac_of_peak": 0.72, "frac_of_ceiling": 3.0 * tf / BF16_MFMA_PEAK_TFLOPS / 0.72,
                                           "source": "profiles/r05_mfma_sustained.txt (tools/ubench/mfmapower.py, 16x16x32 bf16, changing random "
                                                     "operands, 2 waves / SIMD; measured once per round, not by this run)"},
                "hbm_view": {"algorithmic_gbs": gbs, "peak_gbs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}, **common}
    if name.startswith("conv_tangent") or name.startswith("mlp_") or name == "gram_cholesky":
        traffic, note = pmc_traffic("f32", B) if (name == "conv_tangent_t9_ci64_co64" and pmc) else (None, None)
        return {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": note, "kernel": f"{name} (fp32 MFMA)",
                "hbm_view": {"algorithmic_gbs": gbs, "peak_gbs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}, **common}
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "kernel": name, **common}


def end_to_end(config, evals_per_s, precision, density):
    """SURVEY 8d "Which roofline -- End-to-end: MFMA": achieved = W_alg x evals/s with the DENSE algorithmic count W_alg (unchanged
    by the work the decode sweep skips: the structurally-zero coupler's tangent network and the dead checkerboard pixels of the last
    hidden conv -- ``skipped_of_dense_hidden_convs`` says how much of the dominant stage that is), against the fp32-matrix peak
    and, with the split-precision tangent convs, against the bf16 peak in bf16 products (3 per fp32-grade product)."""
    tf = W_ALG_FLOP[config] * evals_per_s / 1e12
    out = {"W_alg_gflop_per_eval": W_ALG_FLOP[config] / 1e9, "W_alg_tflop_per_s": tf, "count": "dense (SURVEY 8d), not reduced by skipped work",
           "fp32_matrix": {"achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS}}
    if precision == "bf16x3" and config in ("c3", "c5"):
        out.update(achieved=3.0 * tf, peak=BF16_MFMA_PEAK_TFLOPS, unit="TFLOP/s (bf16 products, 3 per fp32-grade product)",
                   frac=3.0 * tf / BF16_MFMA_PEAK_TFLOPS)
    else:
        out.update(achieved=tf, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_MFMA_PEAK_TFLOPS)
    try:
        from cmf_amd.bijections import AffineCouplingBijection
        heads = [m for m in density.modules() if type(m).__name__ == "NonSquareHeadDensity"]
        prog = heads[0].program
        acls = [m for m in prog.layers if isinstance(m, AffineCouplingBijection) and m.net.kind == "resnet"]
        if acls:
            per = [2 * sum(1 for b in m.net.module if hasattr(b, "conv1")) for m in acls]            # hidden convs per coupler
            zero = {i for i in prog.zero_in}
            skipped = sum(float(per[k]) if prog._zero(i) else (0.5 if getattr(m, "_live", None) else 0.0)
                          for k, (i, m) in enumerate((i, m) for i, m in enumerate(prog.layers) if m in acls))
            out["skipped_of_dense_hidden_convs"] = {"launch_equivalents": skipped, "of": sum(per), "zero_input_couplers": len(zero)}
            dev = next(density.parameters()).device
            seed = prog._seed_columns(dev)                   # the first decoded coupler runs on its live seed columns only
            if seed is not None:
                nc = (prog.d + 15) // 16 * 16
                k = [i for i, m in enumerate(prog.layers) if m in acls].index(seed["index"])
                full = per[k] - (0.5 if getattr(prog.layers[seed["index"]], "_live", None) else 0.0)
                out["skipped_of_dense_hidden_convs"]["launch_equivalents"] += full * (1.0 - seed["nc"] / nc)
                out["skipped_of_dense_hidden_convs"]["first_coupler_columns"] = {"live": seed["n"], "slots": seed["nc"], "of": nc}
    except Exception:                                       # noqa: BLE001 -- a note, never the measurement
        pass
    return out


def set_kernels(density, tangent=None, primal=None):
    """Arithmetic of every non-square head below ``density`` (``head.kernels``, engine.KernelConfig); None keeps a field."""
    from cmf_amd import engine as E
    for m in density.modules():
        if isinstance(getattr(m, "kernels", None), E.KernelConfig):
            m.kernels = E.KernelConfig(tangent or m.kernels.tangent, primal or m.kernels.primal)


class Workload:
    """One BASELINE configuration on this rank: model, synthetic shard, elbo kwargs."""

    def __init__(self, config, B, rank, device, tangent=None, primal=None):
        self.config = config
        self.dataset, over, _, self.off, self.label = CONFIGS[config]
        self.cfg, self.schema, self.shape, self.sd, self.density = make_model(device, dataset=self.dataset, overrides=over)
        set_kernels(self.density, tangent, primal)
        dequant = self.schema[0]["type"] == "dequantization"
        self.inner = self.density.module.density if dequant else self.density     # noise is part of the synthetic input
        self.B, self.rank, self.device = B, rank, device
        self.x = synth_batch(self.dataset, self.shape, B, rank, device)
        self.kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=self.off, likelihood_wt=1., metric_wt=1.)


def eval_timed(wl, world, steps, warmup, timer_select=None, graph=None, grouped=None):
    """W untimed + K timed ``elbo`` steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, last loss,
    per-kernel HIP-event rows or None, world size the process group reported)."""
    import torch
    import torch.distributed as dist
    from cmf_amd import engine as E
    from cmf_amd.distributed import allreduce_mean_elbo

    def step():
        out = graph(wl.x) if graph is not None else wl.inner.elbo(wl.x, **wl.kw)
        return allreduce_mean_elbo(out["elbo"])          # (sum, count) all-reduce; plain mean on one rank

    grouped = world > 1 if grouped is None else grouped

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    import contextlib
    # per-step durations ("median of >= 10 after 3 warm-ups"): one HIP event on the launch stream after every step, read
    # after the closing fence -- `value` stays on the total wall time between the fences (the driver's own clock sees that)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    with torch.no_grad():
        for _ in range(warmup):
            step()
        with (E.timing(timer_select) if timer_select is not None else contextlib.nullcontext()) as timer:
            fence()
            t0 = time.perf_counter()
            marks[0].record()
            for i in range(steps):
                loss = step()
                marks[i + 1].record()
            fence()
            dt = time.perf_counter() - t0
    wl.step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    rows = timer.by_name() if timer is not None else None
    tmax = torch.tensor([dt], device=wl.device, dtype=torch.float64)
    seen = 1
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        seen = dist.get_world_size()
    return float(tmax.item()), loss, rows, seen


def capture_graph(wl):
    """ElboGraph of the workload's step, or (None, reason) when the capture fails (the run then times eager launches)."""
    import torch
    from cmf_amd.graphs import ElboGraph
    try:
        with torch.no_grad():
            return ElboGraph(wl.inner, wl.x, **wl.kw), None
    except Exception as e:                                  # noqa: BLE001
        torch.cuda.synchronize()
        print(f"[bench] HIP-graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr, flush=True)
        return None, f"{type(e).__name__}: {e}"


HIDDEN_CONV = "conv_tangent_t9_ci64_co64"        # KernelTimer name of the dominant kernel family of the image configurations


def eval_leg(config, B, rank, world, device, steps, warmup, precision, scaling, graph=False, hutchinson=False, primal=None):
    """A secondary evaluation measurement inside the same process (group): returns the sub-object for rank 0's line."""
    import torch
    wl = Workload(config, B, rank, device, precision, primal)
    if hutchinson:
        wl.density.train()
    g = None
    if graph:
        g, _ = capture_graph(wl)
        if world > 1:
            # a capture that fails on ONE rank must not send the ranks down different paths (the eager roofline steps below contain
            # collectives): everybody replays, or nobody does (as in run_rank)
            import torch.distributed as dist
            ok = torch.tensor([1.0 if g is not None else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0:
                g = None
        graph = g is not None
    select = None if graph else ((lambda n: n == HIDDEN_CONV) if config in ("c3", "c5") else (lambda n: True))
    dt, loss, rows, seen = eval_timed(wl, world, steps, warmup, select, g)
    if graph:                                           # replayed graphs cannot carry events: eager steps for the roofline
        _, _, rows, _ = eval_timed(wl, world, 2, 1, lambda n: True, None)
    out = None
    if rank == 0:
        out = {"metric": METRICS[config] + (" (train-mode Hutchinson S=4 + CG)" if hutchinson else ""),
               "value": B * world * steps / dt, "unit": "evals/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
               "per_gpu_batch": B, "global_batch": B * world, "n_gpus": world, "ranks_seen": seen, "scaling": scaling,
               "config": {"workload": wl.label + (", HIP-graph replay" if graph else ""), "loss_mean": float(loss)}}
        if rows:
            kname, n, ms, fl, by = dominant(rows)
            step_ms = 1e3 * dt if not graph else sum(r[1] for r in rows.values())
            out["roofline"] = roofline_of(kname, n, ms, fl, by, step_ms, precision, B, pmc=(config == "c3"))
            if graph:
                out["roofline"]["note"] = "kernel events from 2 eager steps after the timed graph replays; share_of_step = share of GPU kernel time"
    del wl, g
    torch.cuda.empty_cache()
    return out


def init_group(args, world, device):
    """RCCL (or gloo, rehearsal) process group; an initialisation failure ends THIS rank with its reason on stderr and a
    non-zero code (the launcher then stops the others): nothing is re-executed in a process that has touched the GPU."""
    import torch
    import torch.distributed as dist
    try:
        kw = {}
        rdv = os.environ.get("CMF_RENDEZVOUS_FILE")
        if rdv:                                             # started by this file's own launcher: file-store rendezvous, no TCP port
            kw = dict(init_method="file://" + rdv, rank=int(os.environ["RANK"]), world_size=world)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device, **kw)
        else:
            dist.init_process_group(args.backend, **kw)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)                              # the first collective is where RCCL builds its rings
        torch.cuda.synchronize()
        if int(probe.item()) != world:
            raise RuntimeError(f"all-reduce over {world} ranks returned {probe.item()}")
    except Exception as e:                                  # noqa: BLE001 -- anything here is fatal for the measurement
        print(f"[bench] rank {os.environ.get('RANK', '?')}: {args.backend} process group failed: {type(e).__name__}: {e}",
              file=sys.stderr, flush=True)
        sys.exit(3)


def run_rank(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 or args.force_group:
        # BEFORE the first torch.cuda call: the HIP / HSA runtime reads its environment when it initialises
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:                  # --force-group from a bare shell
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                free = s_.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    device = torch.device("cuda", 0 if args.share_gpu else local)
    torch.cuda.set_device(device)
    if world > 1 or args.force_group:
        init_group(args, world, device)
    grouped = world > 1 or args.force_group

    B = args.batch // world if args.strong else args.batch
    wl = Workload(args.config, B, rank, device, args.precision, args.primal_precision)
    cfg, schema, shape, sd, density, inner, x = wl.cfg, wl.schema, wl.shape, wl.sd, wl.density, wl.inner, wl.x
    dataset, off, label = wl.dataset, wl.off, wl.label
    if args.hutchinson:
        assert args.config == "c5", "--hutchinson is C5's train-mode stochastic log-det"
        density.train()                                      # non_square.py:131-138: train mode selects hutch_with_cg

    if args.train:
        line = train_leg(args.steps, args.warmup, args.config, inner, x, B, rank, world, device, off)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if grouped:
            dist.destroy_process_group()
        return

    graph, graph_note = None, None
    if args.graph:
        graph, graph_note = capture_graph(wl)
        if grouped:
            # a capture that fails on ONE rank must not send the ranks down different paths (different numbers of
            # collectives -> a stall until the watchdog): everybody replays, or nobody does
            ok = torch.tensor([1.0 if graph is not None else 0.0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and graph is not None:
                graph, graph_note = None, "HIP-graph capture failed on another rank: every rank times eager launches"
        args.graph = graph is not None

    # the headline keeps its round-1 definition: only the dominant kernel family carries events inside the timed region
    select = None
    if not args.no_kernel_timer and graph is None:
        select = (lambda name: name == HIDDEN_CONV) if args.config in ("c3", "c5") else (lambda name: True)
    dt, loss, rows, seen = eval_timed(wl, world, args.steps, args.warmup, select, graph, grouped)
    per_step = sorted(wl.step_ms)                            # the headline's own steps (later legs re-use ``wl``)
    if graph is not None and not args.no_kernel_timer:       # replayed graphs cannot carry events: eager steps, outside the timed region
        _, _, rows, _ = eval_timed(wl, world, 3, 1, lambda name: True, None)

    # secondary measurements of the default command (SURVEY 8d "Secondary: train-mode fwd+bwd steps/s; per-stage times"; VERDICT r2
    # items 3 / 4): same process (group), after the headline's timed region, each with its own barrier + synchronize bracket
    default_run = (args.config == "c3" and not args.strong and not args.hutchinson and not args.no_legs and args.precision == "bf16x3")
    legs = {}
    if default_run and world > 1:
        # BASELINE configs[3]: the SAME global batch of 512 sharded over the ranks; configs[4]: CIFAR d=128, 32 samples per GPU
        # small shards are launch-latency-sensitive (64 samples: 700 launches in 55 ms): these legs replay a captured HIP graph
        # (measured on one GPU: 55.0 -> 53.2 ms at 64 samples, 68.9 -> 66.9 ms on the C5 shard; nothing at 512 samples, which stays eager
        # with the dominant kernel's events inside the timed region)
        legs["strong_c3"] = eval_leg("c3", max(1, 512 // world), rank, world, device, args.leg_steps, 1, args.precision, "strong",
                                     graph=True, primal=args.primal_precision)
        legs["c5"] = eval_leg("c5", 32, rank, world, device, args.leg_steps, 1, args.precision, "weak", graph=True, primal=args.primal_precision)
    stages = None
    if rank == 0 and not args.train and not args.no_kernel_timer and not args.hutchinson:
        stages = guarded("stages", lambda: stages_leg(wl, 1e3 * dt / args.steps), 1)
    if default_run and world > 1:
        # the gradient bucket's two reduction shapes, then the reference's actual multi-GPU workload (DataParallel TRAINING, 64
        # samples per GPU = configs[3]'s shard) with the faster one.  Last on this model: the optimiser steps change its weights
        n_par = sum((p.numel() + 3) // 4 * 4 for p in inner.parameters() if p.requires_grad)
        legs["grad_reduce"] = grad_reduce_leg(n_par, rank, world, device)
        x64 = synth_batch(dataset, shape, args.train_batch, rank, device)
        legs["train"] = train_leg(args.leg_steps, 1, "c3", inner, x64, args.train_batch, rank, world, device, off, legs["grad_reduce"]["used"])
    f32 = None
    if default_run and world == 1:
        def f32_exact():
            # the same workload with the hidden tangent convs on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): 3 timed steps
            set_kernels(wl.density, tangent="f32")
            try:
                dt32, _, rows32, _ = eval_timed(wl, 1, 3, 1, None if args.no_kernel_timer else (lambda name: name == HIDDEN_CONV), None)
            finally:
                set_kernels(wl.density, tangent=args.precision)
            out = {"value": 3 * B / dt32, "unit": "evals/s", "steps": 3, "warmup": 1, "ms_per_step": 1e3 * dt32 / 3, "dtype": "f32"}
            if rows32:
                name, n, ms, fl, by = dominant(rows32)
                tf = fl / (ms * 1e-3) / 1e12
                out.update(kernel="conv_tangent_kernel<9,4,7> (fp32 MFMA)", kernel_avg_ms=ms / n, kernel_tflops=tf,
                           peak=FP32_MFMA_PEAK_TFLOPS, frac=tf / FP32_MFMA_PEAK_TFLOPS)
            return out

        def c5_train():
            wl5 = Workload("c5", 32, rank, device, args.precision, args.primal_precision)
            try:
                return train_leg(args.leg_steps, 1, "c5", wl5.inner, wl5.x, 32, rank, world, device, wl5.off)
            finally:
                del wl5
                import gc
                gc.collect()                                 # (module graphs hold reference cycles: free the C5 model NOW)
                torch.cuda.empty_cache()

        def train():
            # last on this model: the optimiser steps change its weights (C3 model, the reference's 64-sample C4 shard)
            x64 = synth_batch(dataset, shape, args.train_batch, rank, device)
            return train_leg(args.leg_steps, 1, "c3", inner, x64, args.train_batch, rank, world, device, off)

        if not args.no_f32_exact:
            f32 = guarded("f32_exact", f32_exact, world)
        legs["c5"] = guarded("c5", lambda: eval_leg("c5", 32, rank, world, device, args.leg_steps, 1, args.precision, "weak", graph=True,
                                                     primal=args.primal_precision), world)
        legs["c2b"] = guarded("c2b", lambda: eval_leg("c2b", 4096, rank, world, device, 20, 2, args.precision, "weak", graph=True), world)
        legs["c5_train"] = guarded("c5_train", c5_train, world)
        legs["train"] = guarded("train", train, world)

    if rank == 0:
        total = B * world * args.steps
        d = cfg["latent_dimension"]
        line = {
            "metric": METRICS[args.config] + (" (train-mode Hutchinson S=4 + CG)" if args.hutchinson else ""),
            "value": total / dt, "unit": "evals/s", "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" or args.config not in ("c3", "c5")
                     else "f32 (3x3 tangent convs as bf16x3 split MFMA, fp32 accumulate)",
            "data": "synthetic" + (" (REHEARSAL: ranks share one GPU, not a measurement)" if args.share_gpu else ""),
            "config": {"workload": label + (", HIP-graph replay" if args.graph else ""),
                       "per_gpu_batch": B, "global_batch": B * world, "D": int(torch.tensor(shape).prod()), "latent_dimension": d,
                       "parallelism": f"dp{world}", "loss_mean": float(loss)},
        }
        if rows:
            name, n, ms, fl, by = dominant(rows)
            step_ms = 1e3 * dt if not args.graph else sum(r[1] for r in rows.values())
            line["roofline"] = roofline_of(name, n, ms, fl, by, step_ms, args.precision, B, pmc=(args.config == "c3"))
            if args.graph:
                line["roofline"]["note"] = "kernel events from 3 eager steps after the timed graph replays; share_of_step = share of GPU kernel time"
        n_ = len(per_step)
        line["ms_per_step_median"] = per_step[n_ // 2] if n_ % 2 else 0.5 * (per_step[n_ // 2 - 1] + per_step[n_ // 2])
        line["ms_per_step_minmax"] = [per_step[0], per_step[-1]]
        line["end_to_end"] = end_to_end(args.config, B * world * args.steps / dt, args.precision, inner)
        if stages is not None:
            line["stages"] = stages
        if args.force_group:
            line["config"]["process_group"] = f"{args.backend}, {seen} rank(s), forced"
        if graph_note:
            line["config"]["graph_capture_failed"] = graph_note
        if f32 is not None:
            line["f32_exact"] = f32
        for k, v in legs.items():
            if v is not None:
                line[k] = v
        if world == 1 and args.cpu_batch > 0:
            line["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(schema, shape, {k: v.cpu() for k, v in sd.items()},
                                                                                args.cpu_batch, dataset, off, label), 1)
        line["printed_at_unix"] = time.time()               # (a launcher test bounds the teardown: process-group destruction, exit)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
