"""Round-2 GPU tests: ill-conditioned and retrying reference vectors, training robustness (stale weight packs, jitter under
training, the metric term on the Hutchinson product), the nested prior-dict, the self-launching benchmark.  Every call goes
through the C ABI of libcmf_amd.so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import COND, ROOT, golden_model, load_golden
from test_gpu_parity import build, find_head, inner, rel

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------------------------------------------------
# the reference's jitter loop
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("tag,attempts", [("a", 2), ("b", 4)])
def test_cholesky_retries_match_the_reference_loop(tag, attempts):
    """jitter_retry.npz: Jacobians on which the REFERENCE's loop (non_square.py:262-296) takes 2 and 4 attempts (exactly
    singular in exact arithmetic; the second case has jitters absorbed by fp32 rounding).  The fused Gram + Cholesky kernel
    and its device-side retry chain take the same number of attempts, jitter every sample, and return the reference's
    log-dets and jittered matrices."""
    from cmf_amd import engine as E
    g, _ = load_golden("jitter_retry")
    J = g[f"J_{tag}"].cuda()
    B, D, d = J.shape
    T = E.Tangent.from_dense(J, E.ceil16(d), "panel")
    r = E.gram_cholesky(T, d)
    fail = r.fail.tolist()
    n = 1
    while fail[n - 1]:
        n += 1
    assert n == attempts == int(g[f"attempts_{tag}"])
    assert rel(r.jtj, g[f"jittered_{tag}"]) < 1e-6                     # the jitter lands on EVERY sample's diagonal
    want = g[f"logdet_{tag}"].flatten()
    got = r.logdet.cpu()
    for b in (0, 2):                                                   # well-conditioned samples
        assert abs(float(got[b] - want[b])) / abs(float(want[b])) < 1e-4
    if tag == "a":                                                     # singular sample: pivot ~ 2 eps, resolved to ~4e-6
        assert abs(float(got[1] - want[1])) / abs(float(want[1])) < 1e-4
    else:                                                              # pivot = a few ulp(1024): rounding decides its value
        assert abs(float(got[1] - want[1])) < 1.5
    assert rel(r.l1_diag, torch.diagonal(g[f"jittered_{tag}"], dim1=1, dim2=2).abs().sum(1)) < 1e-5


def _oracle_objective(sd64, ops, x, jitter, kw):
    """-elbo.mean() through the float64 oracle with ``jitter`` added to the diagonal of J^T J before the factorisation --
    what the reference differentiates when its loop retries (the jitter is a constant, non_square.py:284-296)."""
    from oracle import cmf_oracle as O
    pre, head, flow_ops, base, prior_ops = O.split_ops(ops)
    y, lj_pre = O.prehead(pre, x, torch.zeros_like(x))
    z_low, low_elbo, _ = O.encode(sd64, flow_ops, base, prior_ops, y)
    jtj, xh, J = O.jtj_batched(sd64, flow_ops, base, z_low)
    jtj = jtj + jitter * torch.eye(jtj.shape[1], dtype=jtj.dtype)
    logdet = 2 * torch.log(torch.diagonal(torch.linalg.cholesky(jtj), dim1=-2, dim2=-1)).sum(1, keepdim=True)
    l1 = O.metric_l1(jtj, kw.get("add_diagonal_metric_reg", False)) if (kw.get("add_diagonal_metric_reg") or
                                                                          kw.get("add_offdiagonal_metric_reg")) else 0
    recon = ((xh - y).flatten(1) ** 2).sum(-1, keepdim=True) if kw.get("add_reconstruction", True) else 0
    elbo = (low_elbo - logdet / 2) - head["regularization_param"] * recon - l1 + lj_pre
    return -elbo.mean(), y, lj_pre


@pytest.mark.parametrize("kw", [dict(add_offdiagonal_metric_reg=True, add_reconstruction=False),
                                dict(add_diagonal_metric_reg=True, add_reconstruction=False)])
def test_training_differentiates_through_the_jittered_matrix(kw, monkeypatch):
    """A batch whose first factorisation fails trains on (the reference calls its loop with create_graph=self.training): the
    gradient flows through J^T J + eps I.  The failure is forced (flag raised after attempt 0) with a LARGE eps0 so that the
    jitter visibly changes the gradient: HIP gradients = autograd through the float64 oracle with the same jitter, and differ
    from the unjittered gradients."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build("mini_mnist")
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    named = dict(dens.named_parameters())
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    x = g["x"][:3].double()
    eps0 = 0.25                                      # diag(J^T J) is 1 .. 2.6 here; no reconstruction term to drown the log-det's gradient
    orig = E.gram_cholesky

    def forced(T, d, max_attempts=6, eps0_=1e-6):
        r = orig(T, d, 1)
        r.fail[0] = 1
        E.cholesky_retries(r, d, max_attempts, eps0)
        return r

    grads = {}
    for jitter in (eps0, 0.0):
        sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
        loss, y, lj_pre = _oracle_objective(sd64, ops, x, jitter, kw)
        grads[jitter] = (loss.detach(), torch.autograd.grad(loss, [sd64[k] for k in keys], allow_unused=True))
    monkeypatch.setattr(E, "gram_cholesky", forced)
    head.check_cholesky = "lazy"
    pre = lj_pre.float().reshape(-1).cuda()
    loss, elbo, got = head.loss_and_gradients(y.float().cuda(), pre_logjac=pre, **kw)
    assert head.last_gram.fail.tolist()[:2] == [1, 0]                   # exactly one retry ran
    want_loss, want = grads[eps0]
    assert rel(loss, want_loss) < 1e-5
    moved = 0
    for k, w, w0 in zip(keys, want, grads[0.0][1]):
        if w is None or float(w.abs().max()) == 0:
            continue
        assert rel(got[named[k]], w.reshape(named[k].shape)) < 1e-4, k
        moved += rel(w0, w) > 1e-3
    assert moved >= 10                                                   # the jitter is really in the gradient


def test_report_of_attempts_under_training(capsys, monkeypatch):
    """check_cholesky = "sync": the training forward prints the reference's WARNING instead of raising."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build("mini_mnist")
    head = find_head(dens)
    orig = E.gram_cholesky

    def forced(T, d, max_attempts=6, eps0=1e-6):
        r = orig(T, d, 1)
        r.fail[0] = 1
        E.cholesky_retries(r, d, max_attempts, eps0)
        return r

    monkeypatch.setattr(E, "gram_cholesky", forced)
    dens.train()
    with torch.enable_grad():
        out = inner(dens, True).elbo(g["x"].float().cuda(), add_offdiagonal_metric_reg=True)
        (-out["elbo"].mean()).backward()
    assert "2 attempts needed" in capsys.readouterr().out
    assert set(out["prior-dict"]) == {"elbo", "low-dim-x"}              # the training path returns the no-grad path's keys
    with pytest.raises(RuntimeError, match="second time"):
        (-out["elbo"].mean()).backward()


# ----------------------------------------------------------------------------------------------------------------------
# FlatOptimizer: updated weights must reach the kernels
# ----------------------------------------------------------------------------------------------------------------------


def test_flat_optimizer_steps_reach_the_conv_kernels():
    """The fused step writes the flat buffer with a raw kernel: no tensor version moves.  After two steps the model must
    evaluate like a FRESH model built from its state_dict() (the packed weight copies were stale cache hits in round 1), and
    three FlatOptimizer steps must equal three torch.optim.Adam steps on a twin."""
    import cmf_amd
    from cmf_amd.optim import FlatOptimizer
    from cmf_amd.training import train_batch
    g, meta, cfg, dens = build("mini_mnist")
    _, _, _, twin = build("mini_mnist")
    d1, d2 = inner(dens, True), inner(twin, True)
    tcfg = dict(cfg, g_ij_loss=True, g_kk_loss=False)
    train_metrics, _, _ = cmf_amd.get_non_square_train_metrics(tcfg)
    x0 = g["x"].float().cuda()
    kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=True)
    with torch.no_grad():
        d1.eval()
        before = d1.elbo(x0.clone(), **kw)["elbo"].clone()
    flat = FlatOptimizer(d1.parameters(), opt="adam", lr=3e-3)
    ref = torch.optim.Adam(d2.parameters(), lr=3e-3)
    l1, l2 = [], []
    for it in range(3):
        l1.append(float(train_batch(d1, x0.clone(), 10_000, train_metrics, [flat])["metrics"]["loss"].detach()))
        l2.append(float(train_batch(d2, x0.clone(), 10_000, train_metrics, [ref])["metrics"]["loss"].detach()))
    assert np.allclose(l1, l2, rtol=2e-5), (l1, l2)                      # identical trajectories, step by step
    assert l1[1] != l1[0]
    for (k, p), (_, q) in zip(d1.named_parameters(), d2.named_parameters()):
        assert rel(p, q) < 2e-4, k
    d1.eval()
    with torch.no_grad():
        after = d1.elbo(x0.clone(), **kw)["elbo"]
        fresh = cmf_amd.get_density(cmf_amd.get_schema(cfg), g["x"])
        fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in dens.state_dict().items()})
        fresh = fresh.cuda().eval()
        want = inner(fresh, True).elbo(x0.clone(), **kw)["elbo"]
    assert rel(after, want) < 1e-6                                       # the kernels saw the updated conv weights
    assert rel(after, before) > 1e-4                                     # and the parameters did move


def test_weighted_gradient_allreduce_single_rank_is_identity():
    from cmf_amd.optim import FlatOptimizer
    g, meta, cfg, dens = build("c1_sphere")
    opt = FlatOptimizer(dens.parameters(), opt="sgd", lr=1e-2)
    opt.grad.normal_()
    before = opt.grad.clone()
    opt.allreduce_flat(n_local=7)
    assert torch.equal(opt.grad, before)


# ----------------------------------------------------------------------------------------------------------------------
# nested prior-dict (opt-in)
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("name", ["c1_sphere", "mini_mnist", "mini_cifar", "c2b_hepmass"])
def test_nested_prior_dict_matches_the_reference_chain(name):
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    head.nested_prior_dict = True
    with torch.no_grad():
        out = head.elbo(g["head_input"].cuda(), add_offdiagonal_metric_reg=True)
    assert rel(out["elbo"] + g["prehead_logjac"].cuda(), g["elbo_0"]) < 1e-4
    node, level = out["prior-dict"], 0
    while isinstance(node, dict):
        assert sorted(node.keys()) == meta["nested_keys"][level], (level, sorted(node.keys()))
        assert rel(node["elbo"], g["nested_elbo"][level]) < 2e-5, level
        if "low-dim-x" in node:
            assert rel(node["low-dim-x"], g["z_low"]) < 1e-5
        node = node.get("prior-dict")
        level += 1
    assert level == g["nested_elbo"].shape[0]
    head.nested_prior_dict = False
    with torch.no_grad():
        flat = head.elbo(g["head_input"].cuda(), add_offdiagonal_metric_reg=True)
    assert set(flat["prior-dict"]) == {"elbo", "low-dim-x"} and torch.equal(flat["elbo"], out["elbo"])


# ----------------------------------------------------------------------------------------------------------------------
# a14: metric term on the Hutchinson product (S == d)
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("name,diag", [("mini_mnist", False), ("mini_cifar", True), ("c2b_hepmass", False)])
def test_hutchinson_metric_term_and_its_gradients(name, diag):
    """num_hutchinson_samples == latent_dimension: the reference's train-mode objective adds the L1 of the off-diagonal (or
    diagonal) entries of W = (J^T J) eps (non_square.py:87-100 on the third return value of :253-258).  Forward values
    against the float64 oracle; parameter gradients against autograd through it with u = solve(J^T J, eps).detach()."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    named = dict(dens.named_parameters())
    B, d = min(3, g["z_low"].shape[0]), head.program.d
    S = d
    head.log_jacobian_method, head.num_hutchinson_samples, head.max_cg_iterations, head.cg_tolerance = "hutch_with_cg", S, 4 * d, 1e-7
    gen = torch.Generator().manual_seed(78)
    z_low, eps = g["z_low"][:B].float(), torch.randn(B, d, S, generator=gen)
    a, c = torch.randn(B, generator=gen), torch.rand(B, generator=gen) + 0.5
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    jtj, xh, J = O.jtj_batched(sd64, flow_ops, base, z_low.double())
    w = torch.bmm(jtj, eps.double())
    u = torch.linalg.solve(jtj, eps.double()).detach()
    value = (u * w).sum(1).mean(1)
    l1 = O.metric_l1(w, diag).flatten()
    want = torch.autograd.grad((a.double() * value + c.double() * l1).sum(), [sd64[k] for k in keys], allow_unused=True)
    st = head.head_terms_forward(z_low.cuda(), tangents=True, hutch_eps=eps.cuda())
    assert rel(st["hutch"]["l1_diag" if diag else "l1_off"], l1) < 1e-4
    out = head.head_terms_backward(z_low.cuda(), None, g_logdet=a.cuda(), state=st,
                                   **{"g_l1diag" if diag else "g_l1off": c.cuda()})
    checked = 0
    for k, wv in zip(keys, want):
        if wv is not None and float(wv.abs().max()) > 0:
            assert rel(out["grads"][named[k]], wv.reshape(named[k].shape)) < 2e-3, k
            checked += 1
    assert checked >= (40 if len(x_shape) == 3 else 10)
    # through the public API, train mode: elbo = (low - value / 2) - lam rec - wm l1(W)
    dens.train()
    x = g["x"][:B].float().cuda()
    kw = {"add_diagonal_metric_reg" if diag else "add_offdiagonal_metric_reg": True}
    with torch.no_grad():
        torch.manual_seed(5)
        got = inner(dens, "noise" in g).elbo(x.clone(), metric_wt=0.7, **kw)["elbo"]
        h = head.last_hutchinson
        torch.manual_seed(5)
        base_elbo = inner(dens, "noise" in g).elbo(x.clone(), metric_wt=0.7)["elbo"]
    l1_api = O.metric_l1(h["w"].cpu().double(), diag)
    assert rel(got, base_elbo.cpu().double() - 0.7 * l1_api) < 1e-5
    head.num_hutchinson_samples = max(1, d - 1)
    if diag:        # torch.diagonal of the rectangular (d, S) product is valid in the reference (non_square.py:87-92): round 3
        with torch.no_grad():
            assert bool(torch.isfinite(inner(dens, "noise" in g).elbo(x.clone(), **kw)["elbo"]).all())
    else:           # the off-diagonal view(B, d (d - 1)) is not (non_square.py:98)
        with pytest.raises(ValueError, match="num_hutchinson_samples"):
            inner(dens, "noise" in g).elbo(x.clone(), **kw)


def test_hutchinson_cg_probe_chunks():
    """S > 16 probes run in chunks of 16 (one workgroup per sample and chunk): u, w and the value equal the exact solve."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(3)
    for B, d, S in ((5, 64, 64), (3, 128, 128), (4, 21, 40)):
        A = torch.randn(B, d + 8, d, generator=gen, dtype=torch.float64)
        G = torch.bmm(A.transpose(1, 2), A) / d + 0.5 * torch.eye(d, dtype=torch.float64)
        eps = torch.randn(B, d, S, generator=gen, dtype=torch.float64)
        val, u, w, iters = E.hutch_cg(G.float().cuda(), eps.float().cuda(), 4 * d, 1e-7)
        assert rel(w, torch.bmm(G, eps)) < 1e-5
        assert rel(u, torch.linalg.solve(G, eps)) < 1e-3
        assert rel(val, (eps ** 2).sum(1).mean(1)) < 1e-3
        assert int(iters.min()) >= 1


# ----------------------------------------------------------------------------------------------------------------------
# bench.py launches its own ranks
# ----------------------------------------------------------------------------------------------------------------------


def _bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_launches_two_ranks_from_a_bare_shell():
    """``python bench.py --gpus 2`` with no launcher around it: the parent spawns the ranks (gloo + one shared GPU here: a
    rehearsal of the RCCL launch on a one-GPU box), relays ONE JSON line, world size as the process group saw it."""
    line = _bench("--gpus", "2", "--backend", "gloo", "--share-gpu", "--batch", "32", "--steps", "1", "--warmup", "1",
                  "--cpu-batch", "0", "--leg-steps", "1")        # (the training leg at the reference's 64 samples per rank)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
    assert line["value"] > 0 and "REHEARSAL" in line["data"]
    assert "cpu_baseline" not in line and "f32_exact" not in line and "c5_train" not in line
    # round 3: the N > 1 line also carries BASELINE configs[3] (the global batch of 512 sharded over the ranks) and configs[4]
    # (CIFAR d = 128, 32 samples per GPU), timed in the same process group
    s3, c5 = line["strong_c3"], line["c5"]
    for leg in (s3, c5):
        for k in ("metric", "value", "unit", "ms_per_step", "steps", "per_gpu_batch", "global_batch", "ranks_seen", "scaling", "roofline"):
            assert k in leg, k
        assert leg["ranks_seen"] == 2 and leg["value"] > 0
        assert abs(leg["value"] - leg["global_batch"] * leg["steps"] / (leg["ms_per_step"] * 1e-3 * leg["steps"])) / leg["value"] < 1e-6
    assert s3["scaling"] == "strong" and s3["per_gpu_batch"] == 256 and s3["global_batch"] == 512 and "MNIST" in s3["metric"]
    assert c5["scaling"] == "weak" and c5["per_gpu_batch"] == 32 and c5["global_batch"] == 64 and "CIFAR" in c5["metric"]
    # round 4: the gradient bucket summed in both shapes (all-reduce; reduce-scatter + all-gather), and the data-parallel TRAINING
    # step -- what the reference's DataParallel is for -- on the faster one
    gr, tr = line["grad_reduce"], line["train"]
    assert gr["used"] in ("all_reduce", "rs_ag") and gr["all_reduce"] > 0 and gr["rs_ag"] > 0 and gr["bytes"] > 20e6
    assert gr["used"] == min(("all_reduce", "rs_ag"), key=lambda k: gr[k])
    assert tr["n_gpus"] == 2 and tr["config"]["global_batch"] == 128 and tr["config"]["gradient_reduction"] == gr["used"] and tr["value"] > 0


@pytest.mark.parametrize("config", ["c1", "c2b"])
def test_bench_other_configs_print_the_contract(config):
    line = _bench("--config", config, "--steps", "3", "--cpu-batch", "64")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"


# ----------------------------------------------------------------------------------------------------------------------
# two devices in one process (the threading contract of SURVEY 8b)
# ----------------------------------------------------------------------------------------------------------------------


def test_two_devices_one_process():
    """Kernel attributes (dynamic LDS limit) are memoised per device (runtime.hip): the second device of a process must get its
    own.  One thread per device, each building its own replica, like ``nn.DataParallel`` (wrapper.py:52-68).  On a one-GPU box the
    same two threads both use device 0 (the per-thread ``set_device`` + replica + evaluation path still runs, concurrently); with
    two or more GPUs they use devices 1 and 0."""
    import threading
    g, meta, cfg, dens0 = build("mini_mnist")
    x = g["x"].float()
    devs = (1, 0) if torch.cuda.device_count() >= 2 else (0, 0)
    outs, errs = {}, []

    def run(slot, dev):
        try:
            torch.cuda.set_device(dev)
            import cmf_amd
            from cmf_amd.recipe import fill_state_dict
            dens = cmf_amd.get_density(cmf_amd.get_schema(cfg), g["x"])
            dens.load_state_dict(fill_state_dict(dens.state_dict(), seed=meta["recipe_seed"]))
            dens = dens.to(f"cuda:{dev}").eval()
            with torch.no_grad(), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                outs[slot] = inner(dens, True).elbo(x.to(f"cuda:{dev}"), add_offdiagonal_metric_reg=True)["elbo"].cpu()
        except Exception as e:                            # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=run, args=(i, d)) for i, d in enumerate(devs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert rel(outs[0], outs[1]) < 1e-6
    with torch.no_grad():
        want = inner(dens0, True).elbo(x.cuda(), add_offdiagonal_metric_reg=True)["elbo"].cpu()
    assert rel(outs[0], want) < 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# f3: NSF prior (parity unpinned; HIP kernels vs the oracle's restatement of the published algorithm)
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("dataset,layers,hidden", [("power", 2, (16, 16)), ("hepmass", 3, (32, 32)), ("miniboone", 1, (8,))])
def test_nsf_prior_elbo_latents_and_samples_match_the_oracle(dataset, layers, hidden):
    """``prior: "nsf"`` (schemas.py:87-103): rand-channel-perm + LULinear + masked autoregressive rational-quadratic spline
    layers under the tail.  elbo, low-dim elbo, earliest latent and fixed samples of the HIP path against the float64 oracle
    restatement (cmf_rq_spline / cmf_lu_weights / cmf_made_mask_weight + the linear layers on cmf_conv_primal)."""
    from test_oracle_golden import nsf_model
    from oracle import cmf_oracle as O
    cfg, schema, shape, dens, sd, sdo, ops = nsf_model(dataset, layers, hidden)
    dens = dens.cuda().eval()
    head = find_head(dens)
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(16, *shape, generator=gen) * 1.5
    if dataset == "sphere":
        x = x / x.norm(dim=1, keepdim=True)
    with torch.no_grad():
        want = O.elbo(sdo, ops, x.double(), add_offdiagonal_metric_reg=True, return_parts=True)
        got = dens.elbo(x.cuda(), add_offdiagonal_metric_reg=True)
        assert rel(got["elbo"], want["elbo"]) < 1e-4
        z_low, low, earliest = head.program.encode(x.cuda())
        assert rel(low.view(-1, 1), want["parts"]["low_dim_elbo"]) < 1e-4
        assert rel(earliest, O.extract_latent(sdo, ops, x.double(), earliest_latent=True)) < 1e-4
        noise = torch.randn(8, cfg["latent_dimension"], generator=gen) * 1.3
        assert rel(dens.fixed_sample(noise.cuda()), O.fixed_sample(sdo, ops, noise.double())) < 1e-4
        head.nested_prior_dict = True
        nested = dens.elbo(x.cuda(), add_offdiagonal_metric_reg=True)["prior-dict"]
        levels = O.nested_elbos(sdo, ops, x.double())
        node, i = nested, 0
        while isinstance(node, dict):
            assert rel(node["elbo"], levels[i]) < 1e-4, i
            node, i = node.get("prior-dict"), i + 1
        assert i == len(levels)


@pytest.mark.parametrize("dataset,layers,hidden,kw", [
    ("power", 2, (16, 16), dict(add_offdiagonal_metric_reg=True)),
    ("hepmass", 2, (32, 32), dict(add_diagonal_metric_reg=True, likelihood_wt=0.7, metric_wt=0.4)),
])
def test_nsf_prior_training_gradients_match_oracle_autograd(dataset, layers, hidden, kw):
    """``loss.backward()`` through a model with the nsf prior: the gradient of every parameter -- LULinear triangles / diagonal /
    bias, the MADE's masked weights, the coupler networks above the tail through z_low -- against torch.autograd through the
    float64 oracle restatement (spline backward by forward-mode duals, MADE / LULinear backward on the conv kernels)."""
    from test_oracle_golden import nsf_model
    from oracle import cmf_oracle as O
    cfg, schema, shape, dens, sd, sdo, ops = nsf_model(dataset, layers, hidden)
    dens = dens.cuda()
    head = find_head(dens)
    named = dict(dens.named_parameters())
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(6, *shape, generator=gen) * 1.2
    keys = [k for k, v in sdo.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sdo.items()}
    want_elbo = O.elbo(sd64, ops, x.double(), **kw)["elbo"]
    want = torch.autograd.grad(-want_elbo.mean(), [sd64[k] for k in keys], allow_unused=True)
    loss, elbo, grads = head.loss_and_gradients(x.cuda(), **kw)
    assert rel(elbo, want_elbo) < 1e-4 and rel(loss, -want_elbo.mean()) < 1e-4
    checked, prior_checked = 0, 0
    for k, w in zip(keys, want):
        p_ = named[k]
        if w is None or float(w.abs().max()) == 0.0:
            assert p_ not in grads or float(grads[p_].abs().max()) < 1e-12, k
            continue
        assert p_ in grads, k
        assert rel(grads[p_], w.reshape(p_.shape)) < 2e-4, (k, rel(grads[p_], w.reshape(p_.shape)))
        checked += 1
        prior_checked += ("linear." in k) or ("autoregressive_net" in k)
    assert checked >= 40 and prior_checked >= 4 * layers + 10
    # masked positions of the MADE weights get exactly zero gradient
    for m in dens.modules():
        if type(m).__name__ == "_MaskedLinear" and m.weight in grads:
            assert float((grads[m.weight] * (1 - m.mask)).abs().max()) == 0.0
    dens.train()
    dens.zero_grad()
    with torch.enable_grad():
        (-dens.elbo(x.cuda(), **kw)["elbo"].mean()).backward()
    for k, w in zip(keys, want):
        if w is not None and float(w.abs().max()) > 0.0:
            assert rel(named[k].grad, w.reshape(named[k].shape)) < 2e-4, k


def test_rq_spline_kernel_inverse_and_tails():
    from cmf_amd import engine as E
    from oracle import cmf_oracle as O
    gen = torch.Generator().manual_seed(4)
    B, D, bins, hidden = 33, 10, 8, 32
    x = torch.randn(B, D, generator=gen) * 2.5
    x[0, 0], x[0, 1], x[1, 0] = 3.0, -3.0, 0.0                       # the tail bound itself and the middle knot region
    params = torch.randn(B, D * (3 * bins - 1), generator=gen) * 2
    lj = torch.zeros(B, device="cuda")
    z = E.rq_spline(x.cuda(), params.cuda(), bins, hidden, 3.0, lj=lj)
    zw, lw = O.rq_spline(x.double(), params.double().view(B, D, -1), hidden, bins, 3.0)
    assert rel(z, zw) < 1e-5 and rel(lj, lw.sum(1)) < 1e-4
    outside = x.abs() > 3
    assert torch.equal(z.cpu()[outside], x[outside]) and bool(outside.any())
    lji = torch.zeros(B, device="cuda")
    xr = E.rq_spline(z, params.cuda(), bins, hidden, 3.0, inverse=True, lj=lji)
    assert rel(xr, x) < 1e-5 and float((lj + lji).abs().max()) < 1e-3


# ----------------------------------------------------------------------------------------------------------------------
# C5 at full size: CIFAR-shaped D = 3072, d = 128, default 64 x 8 ResNet couplers, the per-GPU shard of 32 samples
# ----------------------------------------------------------------------------------------------------------------------


def test_full_size_c5_properties():
    """BASELINE configs[4] at its full size on one GPU's shard (32 of 256 samples), where the reference-generated fixture
    ``c5_cifar_full`` (B = 2; in test_gpu_parity's fixture lists) pins the values and these size-independent properties
    cover the batch:
      * shard invariance of the eval-mode (exact) elbo;
      * the fused kernel's log-det = fp64 slogdet of the Gram matrix it returned (d = 128: the NT = 8 kernel);
      * train mode, ``hutch_with_cg`` with S = 4 probes (the configuration's stochastic log-det): with CG run to convergence
        the surrogate value is mean_s |eps_s|^2 (u = (J^T J)^-1 eps, so u^T (J^T J) eps = eps^T eps), and J^T J eps from the
        kernel equals the returned Gram matrix applied to the probes;
      * the default CG budget (max_cg_iterations = d, tolerance 1: gpytorch's rule stops after 11 iterations) runs and
        returns finite values -- its iterates are parity-unpinned."""
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    cfg = cmf_amd.get_config("cifar10", latent_dimension=128, hutchinson_samples=4)
    assert cfg["log_jacobian_method"] == "hutch_with_cg"
    B, shape = 32, cmf_amd.DATA_SHAPES["cifar10"]
    gen = torch.Generator().manual_seed(77)
    x = torch.randint(0, 256, (B, *shape), generator=gen).float() + torch.rand(B, *shape, generator=gen)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cfg), x[:2])
    dens.load_state_dict(fill_state_dict(dens.state_dict(), seed=0), strict=True)
    dens = dens.cuda().eval()
    core, head = inner(dens, True), find_head(dens)
    xg = x.cuda()
    with torch.no_grad():
        out = core.elbo(xg.clone(), add_reconstruction=True)["elbo"]            # eval mode: the exact path (non_square.py:131-134)
        gram = head.last_gram
        assert out.shape == (B, 1) and bool(torch.isfinite(out).all()) and gram.jtj.shape == (B, 128, 128)
        sign, want_ld = torch.linalg.slogdet(gram.jtj.double())
        assert bool((sign > 0).all()) and rel(gram.logdet.double(), want_ld) < 1e-5
        part = core.elbo(xg[8:24].clone(), add_reconstruction=True)["elbo"]
        assert rel(part, out[8:24]) < 1e-5
        # train mode: Hutchinson + CG
        dens.train()
        head.max_cg_iterations, head.cg_tolerance = 4 * 128, 1e-7
        tr = core.elbo(xg.clone(), add_reconstruction=True)["elbo"]
        h = head.last_hutchinson
        assert h["eps"].shape == (B, 128, 4)
        assert rel(h["w"], torch.bmm(head.last_gram.jtj.double(), h["eps"].double())) < 1e-5
        assert rel(h["value"], (h["eps"].double() ** 2).sum(1).mean(1)) < 2e-3
        assert bool(torch.isfinite(tr).all())
        head.max_cg_iterations, head.cg_tolerance = 128, 1
        tr2 = core.elbo(xg.clone(), add_reconstruction=True)["elbo"]
        assert bool(torch.isfinite(tr2).all()) and int(head.last_hutchinson["iterations"].max()) == 11


# ----------------------------------------------------------------------------------------------------------------------
# fused MLP coupling layers (cmf_mlp_coupler) against the per-layer launches
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("name,B", [("c1_sphere", 37), ("c2a_power", 300), ("c2b_hepmass", 4096), ("c2b_hepmass", 5)])
def test_fused_mlp_coupling_layers_equal_the_per_layer_path(name, B):
    """One persistent launch per coupling layer (weights streamed through LDS, activations in registers, primal in column
    slot 15) against the per-layer cmf_conv_primal / cmf_conv_tangent + coupling kernels: elbo, J^T J, latents, samples;
    ragged batch sizes exercise the tile tails, B = 4096 several tiles per workgroup."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    gen = torch.Generator().manual_seed(8)
    x = torch.randn(B, *g["x"].shape[1:], generator=gen)
    if name == "c1_sphere":
        x = x / x.norm(dim=1, keepdim=True)
    x = x.cuda()
    res = {}
    for fused in (True, False):
        E.FUSED_MLP = fused
        try:
            with torch.no_grad():
                out = dens.elbo(x.clone(), add_offdiagonal_metric_reg=True)
                z_low, low, earliest = head.program.encode(x.clone())
                res[fused] = dict(elbo=out["elbo"].clone(), jtj=head.last_gram.jtj.clone(), z=z_low, low=low, earliest=earliest,
                                  sample=dens.fixed_sample(earliest[:4].clone()))
        finally:
            E.FUSED_MLP = True
    for k in res[True]:
        assert rel(res[True][k], res[False][k]) < 2e-6, k


def test_fused_mlp_coupler_is_what_the_golden_vectors_ran_on():
    """The reference-vector parity tests of the flat models must have gone through the fused kernel: timer names say so."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build("c2b_hepmass")
    with E.timing(lambda name: True) as timer, torch.no_grad():
        out = dens.elbo(g["x"].cuda(), add_offdiagonal_metric_reg=True)
    names = set(timer.by_name())
    assert rel(out["elbo"], g["elbo_0"]) < 1e-4
    assert "mlp_coupler_tangent" in names and "mlp_coupler_primal" in names
    assert not any(n.startswith("conv_tangent") for n in names)


# ----------------------------------------------------------------------------------------------------------------------
# M-flow baseline head (m_flow: True): Jacobian-free likelihood term, detached latent before the prior
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("lw,rec", [(1.0, False), (0.0, True), (1.0, True)])
def test_m_flow_head_trains_like_the_reference_objectives(lw, rec):
    """ManifoldFlowHeadDensity (non_square.py:341-364): elbo = w_L low_dim_elbo - lambda rec with the latent DETACHED before the
    prior (non_square.py:388-389), alternating objectives (non_square_helpers.py:52-66): the likelihood objective trains the prior
    flows only, the reconstruction objective the coupling stack only.  Gradients against autograd through the float64 oracle
    pieces."""
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    from oracle import cmf_oracle as O
    cfg = cmf_amd.get_config("power", m_flow=True)
    schema, shape = cmf_amd.get_schema(cfg), cmf_amd.DATA_SHAPES["power"]
    dens = cmf_amd.get_density(schema, torch.zeros(1, *shape))
    sd = fill_state_dict(dens.state_dict(), seed=0)
    dens.load_state_dict(sd)
    dens = dens.cuda()
    head = [m for m in dens.modules() if type(m).__name__ == "ManifoldFlowHeadDensity"][0]
    named = dict(dens.named_parameters())
    ops = O.compile_schema(schema, shape)
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn(7, *shape, generator=gen)
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.double().clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    xd = x.double()
    z_low, _, _ = O.encode(sd64, flow_ops, base, [], xd)                      # the coupling stack above the tail
    u, lj = z_low.detach(), torch.zeros(7, 1, dtype=torch.float64)            # prior chain on the DETACHED latent
    for op in prior_ops:
        if op["kind"] == "acl":
            u, l = O.acl_x_to_z(sd64, op, u)
            lj = lj + l
        elif op["kind"] == "gaussian":
            lj = lj + O.gaussian_log_prob(u)
    xh = O.flow_forward(sd64, flow_ops, base, z_low)
    recon = ((xh - xd).flatten(1) ** 2).sum(-1, keepdim=True)
    want_elbo = lw * lj - (hd["regularization_param"] * recon if rec else 0)
    want = torch.autograd.grad(-want_elbo.mean(), [sd64[k] for k in keys], allow_unused=True)
    loss, elbo, grads = head.loss_and_gradients(x.cuda(), likelihood_wt=lw, add_reconstruction=rec)
    assert rel(elbo, want_elbo) < 1e-4
    n_prior = n_flow = 0
    for k, w in zip(keys, want):
        p_ = named[k]
        zero = w is None or float(w.abs().max()) == 0.0
        if zero:
            assert p_ not in grads or float(grads[p_].abs().max()) < 1e-10, k
            continue
        assert rel(grads[p_], w.reshape(p_.shape)) < 1e-4, k
        below_tail = k.count("prior.") > 12                                   # flatten + 10 couplings + the tail hang off the head
        n_prior += below_tail
        n_flow += not below_tail
    assert (n_prior > 0) == (lw > 0) and (n_flow > 0) == rec
    with pytest.raises(ValueError, match="M-flow"):
        head.loss_and_gradients(x.cuda(), add_offdiagonal_metric_reg=True)
    groups = cmf_amd.get_non_square_parameters(dens, True)
    assert len(groups) == 2 and sum(1 for _ in groups[0]) > 0 and sum(1 for _ in groups[1]) > 0


# ----------------------------------------------------------------------------------------------------------------------
# shape sweeps of the round-2 kernels (tails, padding, tile boundaries)
# ----------------------------------------------------------------------------------------------------------------------


@pytest.mark.parametrize("N,d,B", [(784, 64, 7), (1, 64, 3), (5, 49, 4), (100, 57, 2), (785, 63, 3), (33, 64, 1), (4099, 50, 2)])
def test_gram_chol64_shapes(N, d, B):
    """The d <= 64 kernel (49 <= d <= 64 -> NC = 64): row counts that are not multiples of the 4-row groups / of the prefetch
    ring, fewer groups than ring slots, latent dimensions below the padded 64, rank-deficient inputs (N < d: pivot failure)."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(17 * N + d)
    J = torch.zeros(B, N, 64)
    J[:, :, :d] = torch.randn(B, N, d, generator=gen)
    T = E.Tangent(B, N, 64, "panel", "cuda", data=J.reshape(-1).cuda())
    gr = E.gram_cholesky(T, d, max_attempts=1)
    G = torch.einsum("bni,bnj->bij", J[:, :, :d].double(), J[:, :, :d].double())
    assert rel(gr.jtj, G) < 1e-5
    diag = torch.diagonal(G, dim1=1, dim2=2).abs().sum(1)
    assert rel(gr.l1_diag, diag) < 1e-5 and rel(gr.l1_off, G.abs().sum((1, 2)) - diag) < 1e-5
    if N >= d + 8:
        assert gr.fail.tolist()[0] == 0 and int(gr.info.abs().max()) == 0
        assert rel(gr.logdet, torch.linalg.slogdet(G)[1]) < 1e-4
    elif N < d:                                        # rank N < d: a pivot must fail (exact zeros or rounding-level values)
        info = gr.info.cpu()
        assert gr.fail.tolist()[0] == 1 and bool((info > 0).all()) and bool(torch.isnan(gr.logdet).all())


@pytest.mark.parametrize("D,hidden,ncols,B", [(6, [128] * 4, 2, 203), (21, [128] * 4, 10, 77), (21, [128] * 4, 15, 40), (9, [100, 70], 4, 31),
                                              (64, [32] * 4, 1, 65), (3, [10, 10], 3, 19), (43, [128, 16, 128], 7, 129), (8, [33], 5, 16)])
def test_mlp_coupler_shape_sweep(D, hidden, ncols, B):
    """cmf_mlp_coupler against the per-layer launches on single coupling layers: odd widths (padded tiles), unequal hidden
    layers, every packing factor of the TANGENT mode (d = 1 .. 15 -> 8 .. 1 samples per tile), ragged batches, both directions
    of the PRIMAL mode with the log-jacobian, the M-split primal kernel (hidden > 32) and the per-wave one."""
    from cmf_amd import engine as E
    from cmf_amd.bijections import AlternatingChannelwiseAffineCouplingBijection
    from cmf_amd.networks import ChunkedSharedCoupler, get_mlp
    torch.manual_seed(100 * D + ncols)
    for reverse in (False, True):
        layer = AlternatingChannelwiseAffineCouplingBijection(
            (D,), lambda npass: ChunkedSharedCoupler(get_mlp(npass, hidden, 2 * (D - npass))), reverse).cuda()
        for p in layer.parameters():
            p.data.mul_(0.5)
        z0 = torch.randn(B, D, device="cuda")
        V = torch.randn(B, D, ncols, device="cuda")
        out = {}
        for fused in (True, False):
            E.FUSED_MLP = fused
            try:
                z = z0.clone()
                T = E.Tangent.from_dense(V, 16, "fmajor")
                layer.decode_(z, T, ncols=ncols)
                ze, lje = z0.clone(), torch.zeros(B, device="cuda")
                layer.encode_(ze, lje)
                zd, ljd = z0.clone(), torch.zeros(B, device="cuda")
                layer.decode_(zd, None, ljd)
                out[fused] = (z, T.to_dense(16).contiguous(), ze, lje, zd, ljd)
            finally:
                E.FUSED_MLP = True
        for a_, b_ in zip(out[True], out[False]):
            assert rel(a_, b_) < 5e-6
        assert float(out[True][1][:, :, ncols:].abs().max()) == 0.0          # padding columns stay zero
        assert rel(out[True][4], out[True][0]) < 1e-6                           # primal of the tangent mode = primal mode


def test_hutch_cg_probe_counts():
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(9)
    for B, d, S in ((3, 10, 1), (2, 64, 17), (2, 128, 33), (4, 5, 5), (2, 100, 16)):
        A = torch.randn(B, d + 4, d, generator=gen, dtype=torch.float64)
        G = torch.bmm(A.transpose(1, 2), A) / d + 0.3 * torch.eye(d, dtype=torch.float64)
        eps = torch.randn(B, d, S, generator=gen, dtype=torch.float64)
        val, u, w, iters = E.hutch_cg(G.float().cuda(), eps.float().cuda(), 4 * d, 1e-7)
        assert rel(w, torch.bmm(G, eps)) < 1e-5 and rel(u, torch.linalg.solve(G, eps)) < 2e-3
        assert rel(val, (u.cpu().double() * w.cpu().double()).sum(1).mean(1)) < 1e-5
        if S == d:
            off, diag = E.hutch_metric(w)
            W = w.cpu().double()
            dg = torch.diagonal(W, dim1=1, dim2=2).abs().sum(1)
            assert rel(diag, dg) < 1e-5 and rel(off, W.abs().sum((1, 2)) - dg) < 1e-5
