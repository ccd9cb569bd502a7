"""Parity of the HIP path (through the C ABI) against the reference-generated golden vectors and the
CPU oracle.  Needs an MI355X: run with ``-m gpu``.

Tolerances: log-prob / likelihood term 1e-4 relative, g_ij loss 1e-4 relative,
reconstruction 1e-5 relative (fp32; the oracle itself agrees with an fp64 evaluation to ~1e-6).
"""
import numpy as np
import pytest
import torch

from conftest import kink_tolerance, load_golden, golden_model

pytestmark = pytest.mark.gpu

from conftest import COND, FULL, SMALL
ALL = SMALL + COND + FULL


def rel(a, b, floor=1e-9):
    """max |a-b| / max(|b|_max, floor): the floor keeps exact-zero quantities (e.g. the reconstruction error
    of the square d == D sphere case, ~1e-14 of round-off in both implementations) comparable."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(floor))


def per_sample_ok(a, b, tol, abs_floor=None):
    """Per-SAMPLE relative agreement (the max-norm ``rel`` lets a sample with a small |value| hide behind the
    largest one): |a_b - b_b| <= tol * max(|b_b|, 1 % of the batch's largest |b|), or the absolute rounding floor where given
    (conftest.fp64_bound: the g_ij sum is formed by cancellation against the diagonal mass)."""
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    bound = tol * b.abs().clamp_min(0.01 * b.abs().max())
    if abs_floor is not None:
        bound = torch.maximum(bound, abs_floor.detach().cpu().double().reshape(-1))
    return bool(((a - b).abs() <= bound).all())


def build(name):
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    g, meta = load_golden(name)
    cfg = cmf_amd.get_config(meta["dataset"], **meta["overrides"])
    dens = cmf_amd.get_density(cmf_amd.get_schema(cfg), g["x"])
    dens.load_state_dict(fill_state_dict(dens.state_dict(), seed=meta["recipe_seed"], gain=meta.get("recipe_gain")), strict=True)
    dens = dens.cuda().eval()
    return g, meta, cfg, dens


def inner(dens, dequant):
    """Skip the DequantizationDensity wrapper so the test controls the noise (x + u fed directly)."""
    return dens.module.density if dequant else dens


def find_head(dens):
    m = dens
    while type(m).__name__ != "NonSquareHeadDensity":
        mods = m._modules
        m = mods.get("module") or mods.get("density") or mods.get("prior")
    return m


@pytest.mark.parametrize("name", ALL)
def test_elbo_matches_reference_vectors(name):
    g, meta, cfg, dens = build(name)
    dequant = "noise" in g
    x = (g["x"] + g["noise"]) if dequant else g["x"]
    with torch.no_grad():
        for i, (lw, mw, rec, off, diag) in enumerate(meta["elbo_combos"]):
            out = inner(dens, dequant).elbo(x.cuda(), likelihood_wt=lw, metric_wt=mw, add_reconstruction=rec,
                                            add_offdiagonal_metric_reg=off, add_diagonal_metric_reg=diag)
            assert out["elbo"].shape == (x.shape[0], 1)
            assert rel(out["elbo"], g[f"elbo_{i}"]) < 1e-4, (name, i)


@pytest.mark.parametrize("name", ALL)
def test_parts_match_reference_vectors(name):
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    y = g["head_input"].cuda()
    with torch.no_grad():
        z_low, low_elbo, earliest = head.program.encode(y)
        assert rel(z_low, g["z_low"]) < 1e-5
        assert rel(low_elbo.view(-1, 1), g["low_dim_elbo"]) < 1e-5
        assert rel(earliest, g["earliest_latent"]) < 1e-5
        # decode side at the REFERENCE's latent: Jacobian, Gram, Cholesky with no encode rounding in between
        from cmf_amd import engine as E
        d = g["jtj"].shape[1]
        x_hat, T = head.program.decode(g["z_low"].cuda(), tangents=True)
        assert rel(x_hat, g["x_hat"]) < 1e-5
        if "J" in g:
            assert rel(T.to_dense(d), g["J"]) < 1e-4
        gf = E.gram_cholesky(T, d)
        assert rel(gf.jtj, g["jtj"]) < 1e-4 and rel(gf.logdet.view(-1, 1), g["logdet"]) < 1e-4
        # end to end (HIP encode -> decode -> Gram -> Cholesky): 1e-4, unless the REFERENCE's own log-det / g_ij of the fixture
        # move by more when its z_low moves by 1e-6 relative (a few ulps, the rounding of any fp32 encode chain) -- the fixture
        # then sits within rounding of a relu kink and the tolerance is 3 x that movement, computed from the reference-generated
        # ``logdet_pert`` / ``l1_off_pert`` vectors (conftest.kink_tolerance).  mini_mnist_cond1e3 does (its third sample: the HIP
        # encode lands on the other side of the kink, z_low agreeing to 6e-6 of max |z| = 4829).
        tol = kink_tolerance(g, 1e-4)
        # 5.6e-3 there (computed; pinned: a regenerated fixture with a larger perturbation jump must not widen it silently -- ADVICE r3),
        # 1e-4 elsewhere; and the kink is ONE sample's: exactly one sample of that fixture moves by more than 1e-4 under the draws
        assert (tol > 1e-4) == (name == "mini_mnist_cond1e3") and tol < 6e-3, (name, tol)
        if name == "mini_mnist_cond1e3":
            ld = g["logdet"].double().reshape(-1)
            moved = ((g["logdet_pert"].double() - ld).abs().max(0).values / ld.abs().max()) > 1e-4
            assert int(moved.sum()) == 1, moved
        x_hat, J = head.jacobian(z_low)
        assert rel(x_hat, g["x_hat"]) < 1e-5
        if "J" in g:
            assert rel(J, g["J"]) < tol
        head.elbo(y, add_offdiagonal_metric_reg=True)
        gr = head.last_gram
        assert rel(gr.jtj, g["jtj"]) < tol
        assert rel(gr.logdet.view(-1, 1), g["logdet"]) < tol             # log-det itself, 1e-4 relative
        assert gr.attempts == 1 and int(gr.info.abs().max()) == 0
        d = g["jtj"].shape[1]
        off = g["jtj"].abs().sum((1, 2)) - torch.diagonal(g["jtj"], dim1=1, dim2=2).abs().sum(1)
        assert rel(gr.l1_off, off) < tol                                  # g_ij loss
        assert rel(gf.l1_off, off) < 1e-4
        # the same per sample: log-det and g_ij of EVERY sample, at the reference's latent and end to end
        gnorm = 2e-6 * g["jtj"].abs().sum((1, 2))                         # float32 Gram rounding floor (conftest.fp64_bound)
        assert per_sample_ok(gf.logdet, g["logdet"], 1e-4) and per_sample_ok(gf.l1_off, off, 1e-4, gnorm)
        assert per_sample_ok(gr.logdet, g["logdet"], tol) and per_sample_ok(gr.l1_off, off, tol, gnorm)
        assert rel(gr.l1_diag, torch.diagonal(g["jtj"], dim1=1, dim2=2).abs().sum(1)) < tol
        # likelihood term = low_dim_elbo - logdet/2 (the "log-prob" of the north star)
        lik = low_elbo.cpu().view(-1, 1) - gr.logdet.cpu().view(-1, 1) / 2
        assert rel(lik, g["low_dim_elbo"] - g["logdet"] / 2) < 1e-4
    if "prehead_logjac" in g and g["prehead_logjac"].abs().max() > 0:
        xin = (g["x"] + g["noise"]).cuda()
        m, lj = dens.module.density, torch.zeros(xin.shape[0], 1, device="cuda")
        while m is not head:
            r = m.bijection.x_to_z(xin)
            xin, lj = r["z"], lj + r["log-jac"]
            m = m.prior
        assert rel(xin, g["head_input"]) < 1e-5 and rel(lj, g["prehead_logjac"]) < 1e-5


@pytest.mark.parametrize("name", SMALL)
def test_ood_latent_sample_api(name):
    g, meta, cfg, dens = build(name)
    dequant = "noise" in g
    x = ((g["x"] + g["noise"]) if dequant else g["x"]).cuda()
    d = inner(dens, dequant)
    with torch.no_grad():
        o = d.ood(x.clone())
        assert set(o) == {"likelihood", "reconstruction-error"}
        assert rel(o["likelihood"], g["ood_likelihood"]) < 1e-4
        assert rel(o["reconstruction-error"], g["ood_recon"]) < 1e-4
        assert rel(d.extract_latent(x.clone(), earliest_latent=False), g["extract_latent"]) < 1e-5
        assert rel(d.extract_latent(x.clone(), earliest_latent=True), g["extract_earliest"]) < 1e-5
        assert rel(dens.fixed_sample(g["sample_noise"].cuda()), g["fixed_sample"]) < 1e-4
        assert rel(dens.fixed_sample()[:4], g["fixed_sample_default"]) < 1e-4
        s = dens.sample(5)
        assert s.shape == (5, *g["x"].shape[1:]) and torch.isfinite(s).all()


@pytest.mark.parametrize("name", ["c1_sphere", "c2b_hepmass", "mini_mnist", "mini_cifar"])
def test_matches_oracle_on_fresh_inputs(name):
    """Different seed than the fixture: HIP path vs the CPU oracle on the same inputs."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    _, schema, x_shape, ops, sd = golden_model(meta)
    gen = torch.Generator().manual_seed(99)
    B = 5
    if len(x_shape) == 3:
        x = torch.randint(0, 256, (B, *x_shape), generator=gen).float() + torch.rand(B, *x_shape, generator=gen)
    else:
        x = torch.randn(B, *x_shape, generator=gen)
    dequant = "noise" in g
    with torch.no_grad():
        want = O.elbo(sd, ops, x, add_offdiagonal_metric_reg=True, noise=torch.zeros_like(x), return_parts=True)
        got = inner(dens, dequant).elbo(x.cuda(), add_offdiagonal_metric_reg=True)
    assert rel(got["elbo"], want["elbo"]) < 1e-4
    head = find_head(dens)
    assert rel(head.last_gram.logdet.view(-1, 1), want["parts"]["logdet"]) < 1e-4


def test_single_direction_jvp_api():
    g, meta, cfg, dens = build("mini_mnist")
    head = find_head(dens)
    z = g["z_low"].cuda()
    v = torch.randn(z.shape, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        xh, jv = head.jvp_forward(z, v)
    want = torch.einsum("bnd,bd->bn", g["J"], v.cpu())
    assert rel(jv.flatten(1), want) < 1e-4
    assert rel(xh, g["x_hat"]) < 1e-5


def test_bijection_protocol_on_gpu():
    """x_to_z / z_to_x / jvp of single layers: round trip, log-jac sign, tangent vs the oracle's acl_jvp."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build("mini_mnist")
    _, schema, x_shape, ops, sd = golden_model(meta)
    pre, _, flow_ops, base, prior_ops = O.split_ops(ops)
    head = find_head(dens)
    x = g["head_input"]
    node, h = head.prior, x
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for op in flow_ops[:5]:                      # 3 checkerboard ACLs, the squeeze, 1 split-channel ACL
            bij = node.bijection
            r = bij.x_to_z(h.cuda())
            back = bij.z_to_x(r["z"])
            assert rel(back["x"], h) < 1e-5
            assert rel(back["log-jac"], -r["log-jac"], floor=1e-3) < 1e-4
            v = torch.randn(r["z"].shape, generator=gen)
            j = bij.jvp(r["z"], v.cuda())
            if op["kind"] == "acl":
                zr, lj = O.acl_x_to_z(sd, op, h)
                assert rel(r["z"], zr) < 1e-5 and rel(r["log-jac"], lj, floor=1e-3) < 1e-4
                xr, jr = O.acl_jvp(sd, op, zr, v)
                assert rel(j["jvp"], jr) < 1e-4 and rel(j["x"], xr) < 1e-5
            else:
                zr = O.squeeze_x_to_z(h, op["factor"])
                assert rel(r["z"], zr) == 0 and rel(j["jvp"], O.squeeze_z_to_x(v, op["factor"])) == 0
            h, node = zr, node.prior


def test_cholesky_retry_whole_batch_jitter():
    """non_square.py:280-288: a singular J^T J anywhere in the batch jitters every sample."""
    from cmf_amd import engine as E
    B, N, d = 3, 10, 4
    gen = torch.Generator().manual_seed(0)
    J = torch.randn(B, N, 16, generator=gen)
    J[:, :, d:] = 0
    J[1, :, 1] = J[1, :, 0]                           # sample 1: two equal columns -> singular Gram
    T = E.Tangent(B, N, 16, "panel", "cuda", data=J.reshape(-1).cuda())
    gr = E.gram_cholesky(T, d)
    fail = gr.fail.tolist()
    attempts = 1 + next(i for i, f in enumerate(fail) if not f)
    assert attempts >= 2
    G = torch.einsum("bni,bnj->bij", J[:, :, :d], J[:, :, :d])
    added = sum(1e-6 * 10 ** k for k in range(attempts - 1))
    want = G + added * torch.eye(d)
    assert torch.allclose(gr.jtj.cpu(), want, rtol=1e-5, atol=1e-6)
    L = torch.linalg.cholesky(want[[0, 2]].double())
    ld = 2 * torch.log(torch.diagonal(L, dim1=1, dim2=2)).sum(1)
    assert torch.allclose(gr.logdet.cpu()[[0, 2]].double(), ld, rtol=1e-4)


def test_cpu_tensors_are_refused():
    g, meta, cfg, dens = build("c1_sphere")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dens.elbo(g["x"])


@pytest.mark.parametrize("name", ["c1_sphere", "c2b_hepmass", "mini_mnist", "mini_cifar"])
def test_hutchinson_surrogate(name):
    """Row a14: J^T J eps is pinned by the reference vectors; the CG iterates follow the build's documented rule
    (gpytorch's solver is un-vendored) and are checked against its CPU restatement and the exact solve."""
    from cmf_amd import engine as E
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    eps = g["hutch_eps"].cuda()
    d = eps.shape[1]
    with torch.no_grad():
        x_hat, T = head.program.decode(g["z_low"].cuda(), tangents=True)
        gr = E.gram_cholesky(T, d)
        val, u, w, iters = E.hutch_cg(gr.jtj, eps, max_iter=4 * d, tol=1e-7, min_iter=1)       # run to convergence
        assert rel(w, g["hutch_jtj_eps"]) < 1e-4                                                # pinned by the reference
        exact = torch.linalg.solve(g["jtj"].double(), g["hutch_eps"].double())
        assert rel(u, exact) < 1e-3
        assert rel(val, (g["hutch_eps"] ** 2).sum(1).mean(1)) < 1e-3
        for max_iter in (1, 2, 5):
            val2, u2, w2, it2 = E.hutch_cg(gr.jtj, eps, max_iter=max_iter, tol=1.0)
            uo, ito = O.cg_documented(g["jtj"], g["hutch_eps"], max_iter, 1.0)
            assert it2.cpu().tolist() == ito.tolist()
            assert rel(u2, uo) < 1e-4


def test_train_mode_hutchinson_elbo():
    g, meta, cfg, dens = build("mini_mnist")
    head = find_head(dens)
    head.log_jacobian_method, head.num_hutchinson_samples, head.max_cg_iterations = "hutch_with_cg", 3, 4
    x = (g["x"] + g["noise"]).cuda()
    with torch.no_grad():
        dens.train()
        out = inner(dens, True).elbo(x, add_reconstruction=True)
        h = head.last_hutchinson
        assert h["w"].shape == (3, 4, 3) and torch.isfinite(out["elbo"]).all()
        want = g["low_dim_elbo"] - h["value"].cpu().view(-1, 1) / 2 - 50 * g["ood_recon"] + g["prehead_logjac"]
        assert rel(out["elbo"], want) < 1e-4
        with pytest.raises(ValueError):
            inner(dens, True).elbo(x, add_offdiagonal_metric_reg=True)
        dens.eval()                                   # eval always takes the exact path (non_square.py:133)
        out = inner(dens, True).elbo(x, add_reconstruction=True, add_offdiagonal_metric_reg=True)
        assert rel(out["elbo"], g["elbo_0"]) < 1e-4


@pytest.mark.parametrize("name", ["c1_sphere", "c2b_hepmass", "mini_mnist"])
def test_hip_graph_replay_matches_eager(name):
    """A captured elbo replays to the same numbers as the eager call, also on new inputs of the same shape."""
    from cmf_amd.graphs import ElboGraph
    g, meta, cfg, dens = build(name)
    dequant = "noise" in g
    d = inner(dens, dequant)
    x = ((g["x"] + g["noise"]) if dequant else g["x"]).cuda()
    eg = ElboGraph(d, x, add_reconstruction=True, add_offdiagonal_metric_reg=True)
    out = eg(x)["elbo"].clone()
    assert rel(out, g["elbo_0"]) < 1e-4
    x2 = torch.flip(x, dims=(0,)).contiguous()
    out2 = eg(x2)["elbo"].clone()
    with torch.no_grad():
        want2 = d.elbo(x2.clone(), add_reconstruction=True, add_offdiagonal_metric_reg=True)["elbo"]
    assert rel(out2, want2) < 1e-6
    assert eg.cholesky_attempts() == 1
    # replays must be idempotent (a captured hipMemsetAsync node once re-armed the jitter retries from replay 2 on)
    for _ in range(4):
        assert torch.equal(eg(x2)["elbo"], out2)
    assert eg.cholesky_attempts() == 1


def test_full_size_c3_properties():
    """BASELINE configs[2] at its FULL size (MNIST-shaped, B = 512, d = 64), where the CPU oracle would need half an
    hour: size-independent properties instead of a reference value.
      * latent round trip: decoding lands on the manifold, so encode(decode(z)) == z;
      * shard invariance: the elbo of a slice of the batch equals the slice of the elbo (what one-process-per-GPU
        sharding relies on);
      * linearity: a single-direction JVP equals the full Jacobian applied to that direction, through the Gram matrix:
        |J v|^2 == v^T (J^T J) v;
      * the fused kernel's log-det agrees with an fp64 slogdet of the Gram matrix it returned, and its off-diagonal L1
        with a recomputation from that matrix."""
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    cfg = cmf_amd.get_config("mnist", latent_dimension=64, g_hidden_channels=[64] * 8, log_jacobian_method="cholesky")
    B, shape = 512, cmf_amd.DATA_SHAPES["mnist"]
    gen = torch.Generator().manual_seed(4321)
    x = torch.randint(0, 256, (B, *shape), generator=gen).float() + torch.rand(B, *shape, generator=gen)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cfg), x[:4])
    dens.load_state_dict(fill_state_dict(dens.state_dict(), seed=0), strict=True)
    dens = dens.cuda().eval()
    core, head = inner(dens, True), find_head(dens)
    xg = x.cuda()
    with torch.no_grad():
        out = core.elbo(xg.clone(), add_reconstruction=True, add_offdiagonal_metric_reg=True)["elbo"]
        gram = head.last_gram
        jtj, logdet, l1 = gram.jtj.double(), gram.logdet.double(), gram.l1_off.double()
        assert out.shape == (B, 1) and bool(torch.isfinite(out).all())
        # fused log-det / L1 vs fp64 recomputation from the returned Gram matrices
        sign, want_ld = torch.linalg.slogdet(jtj)
        assert bool((sign > 0).all()) and rel(logdet, want_ld) < 1e-5
        off = jtj.abs().sum((1, 2)) - jtj.diagonal(dim1=1, dim2=2).abs().sum(1)
        assert rel(l1, off) < 1e-5
        # shard invariance
        part = core.elbo(xg[128:192].clone(), add_reconstruction=True, add_offdiagonal_metric_reg=True)["elbo"]
        assert rel(part, out[128:192]) < 1e-5
        # latent round trip on the manifold (z_low = the latent the head decodes from)
        z = core.extract_latent(xg.clone(), earliest_latent=False)
        xh = head.flow_forward(z)
        z2, _, _ = head.program.encode(xh)
        assert rel(z2, z) < 1e-4
        # linearity: one direction pushed through the decode == the full Jacobian (same z) applied to it.  (Not compared
        # with the elbo call's Gram matrix: that call reaches z through the fused pre-head kernel, one ulp away, and a
        # relu mask that flips on a pre-activation at zero moves J of that sample by ~1e-4 -- in ANY fp32 implementation.)
        v = torch.randn(z.shape, generator=torch.Generator().manual_seed(5)).cuda()
        _, jv = head.jvp_forward(z, v)
        _, J = head.jacobian(z)
        want = torch.einsum("bnd,bd->bn", J.double(), v.double())
        err = (jv.flatten(1).double() - want).norm(dim=1) / want.norm(dim=1)
        assert float(err.max()) < 1e-5


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's contract keys (+ roofline / cpu_baseline objects); run on a small
    batch so that it takes seconds."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--batch", "32", "--cpu-batch", "8", "--leg-steps", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and isinstance(c.get("cpu_model"), str) and c["cpu_model"]
    # round 5 (SURVEY 8d's protocol): a warm-up call, then the cost at B/4, B/2, B; the median step beside the mean; the
    # end-to-end roofline object from the dense algorithmic count W_alg
    assert c["warmup_batch"] >= 1 and set(c["evals_per_s_by_batch"]) == {"2", "4", "8"} and c["value"] == c["evals_per_s_by_batch"]["8"]
    assert d["ms_per_step_median"] > 0 and d["ms_per_step_minmax"][0] <= d["ms_per_step_median"] <= d["ms_per_step_minmax"][1]
    assert abs(d["ms_per_step_median"] - d["ms_per_step"]) < 0.2 * d["ms_per_step"]
    e = d["end_to_end"]
    assert e["W_alg_gflop_per_eval"] == 290.42 and abs(e["W_alg_tflop_per_s"] - 290.42e-3 * d["value"]) < 1e-6 * e["W_alg_tflop_per_s"]
    assert abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-12 and 0 < e["frac"] < 1 and e["peak"] == 2500.0
    # 16 (the zero-input coupler) + 7 x 0.5 (checkerboard tails) + 15.5 x (1 - 48 / 64) (the first coupler's 36 live seed columns)
    assert e["skipped_of_dense_hidden_convs"] == {"launch_equivalents": 23.375, "of": 160, "zero_input_couplers": 1,
                                                  "first_coupler_columns": {"live": 36, "slots": 48, "of": 64}}
    assert "r05_mfma_sustained" in d["roofline"]["live_data_mfma_ceiling"]["source"]
    # round 4 (SURVEY 8d "per-stage times"): stages in ms / step that sum to ms_per_step (host_gap is the remainder)
    st = d["stages"]
    names = ("encode", "primal", "tangent_hidden", "tangent_first_last", "acl", "gram_cholesky", "other", "host_gap")
    for k in names:
        assert k in st, k
    assert abs(sum(st[k] for k in names) - d["ms_per_step"]) <= 0.03 * d["ms_per_step"]
    assert all(st[k] > 0 for k in names[:-2]) and st["tangent_hidden"] > st["primal"] > 0
    assert abs(st["kernels_sum"] - sum(st[k] for k in names[:-1])) < 1e-2
    # round 3: the secondary measurements of SURVEY 8d ride in the default line, on the driver's clock
    for leg, unit, per_gpu in (("train", "samples/s", 64), ("c5", "evals/s", 32), ("c5_train", "samples/s", 32), ("c2b", "evals/s", 4096)):
        assert leg in d, leg
        e = d[leg]
        for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "config"):
            assert k in e, (leg, k)
        assert e["unit"] == unit and e["value"] > 0 and e["steps"] >= 1 and "workload" in e["config"]
        assert (e.get("per_gpu_batch") or e["config"].get("per_gpu_batch")) == per_gpu
    assert d["c5"]["roofline"]["bound"] == "mfma" and 0 < d["c5"]["roofline"]["frac"] < 1 and d["c5"]["roofline"]["traffic"] is None
    assert d["c2b"]["roofline"]["frac"] > 0 and "graph" in d["c2b"]["config"]["workload"]
    assert "graph" in d["c5"]["config"]["workload"] and "eager steps" in d["c5"]["roofline"]["note"]      # round 5: the small shard replays a graph
    assert d["c5_train"]["config"]["peak_memory_gib"] < 30 and "Hutchinson" in d["c5_train"]["config"]["workload"]
    assert "kernel source" in (d["roofline"].get("traffic_source") or "")


@pytest.mark.parametrize("name", ALL)
def test_reverse_sweep_vjp_and_matrix_free_jtj(name):
    """J^T w from the reverse sweep (transposed convs + ACL adjoints) equals the dense Jacobian applied to w, and the
    matrix-free (J^T J) v of non_square.py:190-201 equals the Gram matrix of the exact path applied to v; its value is
    also pinned by the reference vector JtJ.eps of the fixture (hutch_jtj_eps) where present."""
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    z = g["z_low"].cuda()
    B, d = z.shape
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        x_hat, J = head.jacobian(z)                                   # (B, D, d)
        D = J.shape[1]
        w = torch.randn(B, D, 3, generator=gen).cuda()
        xh2, got = head.program.vjp(z, w)
        want = torch.einsum("bnd,bns->bds", J.double(), w.double())
        assert rel(xh2, x_hat) < 1e-6
        assert rel(got, want) < 2e-5
        _, one = head.vjp_forward(z, w[:, :, 0].reshape(B, *x_hat.shape[1:]))
        assert rel(one, want[:, :, 0]) < 2e-5
        v = torch.randn(B, d, 2, generator=gen).cuda()
        _, mv = head.jtj_matvec(z, v)
        G = torch.einsum("bni,bnj->bij", J.double(), J.double())
        assert rel(mv, torch.einsum("bij,bjs->bis", G, v.double())) < 2e-5
        if "hutch_eps" in g and "hutch_jtj_eps" in g:
            _, ref_mv = head.jtj_matvec(z, g["hutch_eps"].cuda())
            assert rel(ref_mv, g["hutch_jtj_eps"]) < 1e-4


# test_parity_statistics_full_mnist_model (32 fresh full-size inputs against the CPU oracle with a typed-in 5e-4 per-sample bound)
# moved to tests/test_gpu_round3.py::test_full_size_statistics_against_the_float64_reference: the same 32 inputs against the
# REFERENCE in float32 and float64 (fixture c3_mnist_stats32), per-sample bounds computed from the reference's own yardsticks.


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar", "c2b_hepmass", "c1_sphere"])
def test_coupler_net_weight_gradients_match_autograd(name):
    """f1 building block: the weight gradients of a coupler network's TANGENT pass -- forward sweep keeping every layer's
    input tangent, reverse sweep (transposed convs) with ``cmf_conv_tangent_wgrad`` per layer -- against torch.autograd
    through a float64 restatement of the same tangent network, for the first and the last coupling layer of the model."""
    import torch.nn.functional as F
    from cmf_amd import engine as E
    from cmf_amd.bijections import AffineCouplingBijection
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    layers = [m for m in head.program.layers if isinstance(m, AffineCouplingBijection)]
    gen = torch.Generator().manual_seed(5)
    B, S, nc = 3, 5, 16
    for bij in (layers[0], layers[-1]):
        geo, net, layout = bij.geom, bij.net, bij.layout
        view = bij.view("cuda")
        with torch.no_grad():
            z = torch.randn(B, *geo.shape, generator=gen).cuda()
            y, _, acts = E.net_primal(net, z, view, need_acts=True)
            t_in = torch.randn(B, geo.N, S, generator=gen)
            c_out = torch.randn(B, y[0].numel(), S, generator=gen)
            T = E.Tangent.from_dense(t_in.cuda(), nc, layout)
            saved, grads = [], {}
            YT = E.net_tangent(net, T, view, acts, save=saved)
            Ct = E.Tangent.from_dense(torch.zeros(B, geo.N, S).cuda(), nc, layout)
            E.net_cotangent(net, E.Tangent.from_dense(c_out.cuda(), nc, layout), view, acts, Ct, saved=saved, grads=grads)
        # float64 restatement of the tangent network with the SAME masks (primal activations are constants here)
        chans = [view.chan_off + i * view.chan_step for i in range(view.cin)]
        acts64 = [a.detach().cpu().double() for a in acts]
        if net.kind == "resnet":
            conv0, blocks, convf = E._resnet_parts(net)
            mods = [conv0] + [c for b in blocks for c in (b.conv1, b.conv2)] + [convf]
            ws = [m.weight.detach().cpu().double().requires_grad_(True) for m in mods]
            H, W = geo.H, geo.W
            conv = lambda x, w: F.conv2d(x.permute(0, 4, 1, 2, 3).reshape(B * S, x.shape[1], H, W), w, padding=w.shape[-1] // 2) \
                .reshape(B, S, w.shape[0], H, W).permute(0, 2, 3, 4, 1)
            r = lambda a: (a > 0).double().unsqueeze(-1)
            x0 = t_in.double().reshape(B, geo.C, H, W, S)[:, chans]
            if view.mask is not None:
                x0 = x0 * view.mask.detach().cpu().double().reshape(1, view.cin, H, W, 1)
            h = conv(x0, ws[0])
            for k in range(len(blocks)):
                u = conv(r(acts64[2 * k]) * h, ws[1 + 2 * k])
                h = conv(r(acts64[2 * k + 1]) * u, ws[2 + 2 * k]) + h
            out = conv(r(acts64[-1]) * h, ws[-1]).reshape(B, -1, S)
        else:
            mods = [m for m in net if isinstance(m, torch.nn.Linear)]
            ws = [m.weight.detach().cpu().double().requires_grad_(True) for m in mods]
            h = t_in.double()[:, chans]                                             # (B, cin, S)
            for i, w in enumerate(ws):
                if i > 0:
                    h = (1 - acts64[i - 1] ** 2).unsqueeze(-1) * h
                h = torch.einsum("oc,bcs->bos", w, h)
            out = h
        assert rel(YT.to_dense(S), out) < 2e-5
        want = torch.autograd.grad((out * c_out.double()).sum(), ws + [])
        assert len(grads) == len(mods)
        for m, gw in zip(mods, want):
            assert rel(grads[m.weight], gw) < 1e-4, (name, type(m).__name__, tuple(m.weight.shape))


@pytest.mark.parametrize("name,B", [("mini_mnist", 5), ("mini_cifar", 16), ("c3_mnist_full", 3)])
def test_resnet_coupler_primal_backward_matches_autograd(name, B):
    """f1 building block: primal backward of a ResNet coupler network (ScaledTanh stage, transposed convs with per-column relu',
    weight / bias gradients, input cotangent) on the tangent-conv kernels with 16 samples in the column slots, against
    torch.autograd through a float64 restatement; first (checkerboard mask) and last (channel split) coupling layer."""
    import torch.nn.functional as F
    from cmf_amd import engine as E
    from cmf_amd.bijections import AffineCouplingBijection
    g_, meta, cfg, dens = build(name)
    head = find_head(dens)
    layers = [m for m in head.program.layers if isinstance(m, AffineCouplingBijection)]
    gen = torch.Generator().manual_seed(6)
    for bij in (layers[0], layers[-1]):
        geo, net = bij.geom, bij.net
        view = bij.view("cuda")
        conv0, blocks, convf = E._resnet_parts(net)
        with torch.no_grad():
            z = torch.randn(B, *geo.shape, generator=gen).cuda()
            y, g, acts = E.net_primal(net, z, view, need_acts=True)
            dy, dg = torch.randn(y.shape, generator=gen), torch.randn(y.shape, generator=gen)
            grads, dz = {}, torch.zeros_like(z)
            E.net_primal_backward(net, z, view, acts, y, g, dy.cuda(), dg.cuda(), grads, dz)
        params = [conv0.weight] + [p for b in blocks for p in (b.conv1.weight, b.conv1.bias, b.conv2.weight, b.conv2.bias)] + \
                 [convf.weight, convf.bias, net.weights, net.bias]
        p64 = [p.detach().cpu().double().requires_grad_(True) for p in params]
        it = iter(p64)
        zd = z.cpu().double().requires_grad_(True)
        chans = [view.chan_off + i * view.chan_step for i in range(view.cin)]
        x0 = zd[:, chans]
        if view.mask is not None:
            x0 = x0 * view.mask.detach().cpu().double().reshape(1, view.cin, geo.H, geo.W)
        a = F.conv2d(x0, next(it), padding=1)
        for _ in blocks:
            c1 = F.conv2d(torch.relu(a), next(it), next(it), padding=1)
            a = a + F.conv2d(torch.relu(c1), next(it), next(it), padding=1)
        u = F.conv2d(torch.relu(a), next(it), next(it))
        sw, sb = next(it), next(it)
        t = torch.tanh(u)
        yy, gg = sw * t + sb, sw * (1 - t * t)
        assert rel(y, yy) < 1e-5 and rel(g, gg) < 1e-4
        want = torch.autograd.grad((yy * dy.double()).sum() + (gg * dg.double()).sum(), p64 + [zd])
        errs = [rel(grads[p], w.reshape(p.shape)) for p, w in zip(params, want[:-1])]
        print(f"{name}: dz err {rel(dz, want[-1]):.1e}, parameter gradient errors max {max(errs):.1e} median {sorted(errs)[len(errs) // 2]:.1e}")
        assert rel(dz, want[-1]) < 1e-4
        for p, w in zip(params, want[:-1]):
            assert rel(grads[p], w.reshape(p.shape)) < 1e-4, (name, tuple(p.shape))


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar", "mini_mnist_small", "c3_mnist_full"])
def test_head_terms_parameter_gradients_match_oracle_autograd(name):
    """f1: d/d theta and d/d z_low of  sum_b a_b logdet(J^T J) + o_b sum_{i!=j}|G_ij| + r_b ||x_hat - x||^2  at fixed z_low --
    tangent sweep with saved state, Gram backward, reverse sweep with weight gradients, coupling-layer cross terms, primal
    backward of the coupler networks -- against torch.autograd through the float64 CPU oracle (the reference's own route:
    loss.backward() through the JVP graphs, trainer.py:213)."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    gen = torch.Generator().manual_seed(41)
    z_low = g["z_low"][:3].float()
    B, d = z_low.shape
    a, o, r = (torch.randn(B, generator=gen) for _ in range(3))
    xt = torch.randn(B, *head.program.tail.x_shape if False else g["x_hat"].shape[1:], generator=gen)
    # oracle, float64, autograd
    keys = [k for k, v in sd.items() if v.is_floating_point() and any(k.endswith(s) for s in (".weight", ".bias", ".weights"))]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    zd = z_low.double().requires_grad_(True)
    jtj, xh, J = O.jtj_batched(sd64, flow_ops, base, zd)
    logdet, jtj2, attempts = O.cholesky_logdet(jtj)
    assert attempts == 1
    loss = (a.double() * logdet.view(-1) + o.double() * O.metric_l1(jtj, False).view(-1)
            + r.double() * ((xh - xt.double()).flatten(1) ** 2).sum(1)).sum()
    used = [k for k in keys]
    want = torch.autograd.grad(loss, [sd64[k] for k in used] + [zd], allow_unused=True)
    out = head.head_terms_backward(z_low.cuda(), xt.cuda(), g_logdet=a.cuda(), g_l1off=o.cuda(), g_rec=r.cuda())
    assert rel(out["x_hat"], xh) < 1e-5 and rel(out["logdet"], logdet.view(-1)) < 1e-4
    named = dict(dens.named_parameters())
    errs = []
    for k, w in zip(used, want[:-1]):
        p = named[k]
        if w is None:                                   # parameters outside the decode path (prior flows)
            assert p not in out["grads"]
            continue
        assert p in out["grads"], k
        errs.append(rel(out["grads"][p], w.reshape(p.shape)))
    print(f"{name}: dz_low err {rel(out['dz_low'], want[-1]):.1e}; {len(errs)} tensors, max {max(errs):.1e} median {sorted(errs)[len(errs) // 2]:.1e}")
    assert rel(out["dz_low"], want[-1]) < 1e-4
    assert max(errs) < 1e-4 and len(errs) >= 30


@pytest.mark.parametrize("name,kw", [
    ("mini_mnist", dict(add_offdiagonal_metric_reg=True)),
    ("mini_cifar", dict(add_diagonal_metric_reg=True, likelihood_wt=0.5, metric_wt=0.3)),
    ("mini_mnist_small", dict(add_reconstruction=False)),
    ("c1_sphere", dict(add_offdiagonal_metric_reg=True)),
    ("c1_sphere_d2", dict(add_offdiagonal_metric_reg=True, likelihood_wt=0.7)),
    ("c2a_power", dict(add_diagonal_metric_reg=True, metric_wt=0.2)),
    ("c2b_hepmass", dict(add_offdiagonal_metric_reg=True)),
])
def test_loss_and_gradients_match_oracle_autograd(name, kw):
    """f1: loss = -elbo.mean() and d loss / d theta for EVERY parameter (coupler networks above the base through encode, decode,
    tangent stack and cross terms; the low-dimensional prior flows) against torch.autograd through the float64 CPU oracle --
    the reference's ``loss.backward()`` (trainer.py:207-215) -- on the image fixtures (ResNet couplers)."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    B = min(4, g["x"].shape[0])
    x = g["x"][:B].double()
    noise = torch.zeros_like(x)
    named = dict(dens.named_parameters())
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    want_elbo = O.elbo(sd64, ops, x, noise=noise, **kw)["elbo"]
    want = torch.autograd.grad(-want_elbo.mean(), [sd64[k] for k in keys], allow_unused=True)
    y, lj_pre = O.prehead(O.split_ops(ops)[0], x, noise)
    pre = None if lj_pre is None or not torch.is_tensor(lj_pre) else lj_pre.float().reshape(-1).cuda()
    loss, elbo, grads = head.loss_and_gradients(y.float().cuda(), pre_logjac=pre, **kw)
    assert rel(elbo, want_elbo) < 1e-4 and rel(loss, -want_elbo.mean()) < 1e-4
    # The full-size model is NOT in this list on purpose.  Its 4.3 M relu pre-activations per sample include some within fp32
    # rounding of zero: the HIP chain's layer inputs differ from the float64 oracle's by ~1e-7 relative, ONE activation of the last
    # encode layer lands on the other side of zero (|a| < 2e-6) and that single mask flip moves every gradient downstream of it by
    # ~5e-4 (tests/dev/encode_chain_check.py: with the oracle's layer input the same step agrees to 9e-8; tests/dev/
    # grad_full_check.py: worst 1.5e-2, median 2.5e-4, identical for the fp32 and the split kernels).  That is the exact gradient
    # of the function fp32 evaluates, not an error of the backward pass -- whose full-size accuracy is pinned flip-free by
    # test_head_terms_parameter_gradients_match_oracle_autograd[c3_mnist_full] (decode side incl. the split forward / reverse /
    # weight-gradient kernels: 1.8e-6) and test_encode_layers_backward_full_size (every encode layer on the oracle's inputs).
    full = name == "c3_mnist_full"
    errs = {}
    for k, w in zip(keys, want):
        p = named[k]
        if w is None or float(w.abs().max()) == 0.0:
            assert p not in grads or float(grads[p].abs().max()) == 0.0, k
            continue
        assert p in grads, k
        errs[k] = rel(grads[p], w.reshape(p.shape))
    worst_key = max(errs, key=errs.get)
    worst, checked = errs[worst_key], len(errs)
    median = float(np.median(list(errs.values())))
    print(f"{name}: {checked} parameter tensors, relative gradient error: worst {worst:.2e} ({worst_key}), median {median:.2e}")
    assert worst < (5e-2 if full else 1e-4), (worst_key, worst)
    assert median < (2e-3 if full else 2e-5)
    assert checked >= (40 if len(x_shape) == 3 else 10)
    # the same through autograd, as the reference trainer drives it: elbo -> loss -> loss.backward() -> p.grad
    dens.train()
    dens.zero_grad()
    with torch.enable_grad():
        out = inner(dens, "noise" in g).elbo(g["x"][:B].float().cuda(), **kw)
        assert out["elbo"].requires_grad and rel(out["elbo"], want_elbo) < 1e-4
        (-out["elbo"].mean()).backward()
    for k, w in zip(keys, want):
        if w is not None and float(w.abs().max()) > 0.0:
            assert rel(named[k].grad, w.reshape(named[k].shape)) < (5e-2 if full else 1e-4), k
        else:
            assert named[k].grad is None or float(named[k].grad.abs().max()) == 0.0


def test_training_steps_reduce_the_loss():
    """Five optimiser steps on a fixed mini-batch through the reference's training closure (``get_non_square_train_metrics``,
    ``loss.backward()``) with the flat-buffer Adam: the loss goes down, and warm-up epochs (likelihood_wt = 0) run too."""
    import cmf_amd
    from cmf_amd.optim import FlatOptimizer
    g, meta, cfg, dens = build("mini_mnist")
    dens = inner(dens, True)                       # no dequantisation noise: the loss is a deterministic function of the parameters
    dens.train()
    cfg = dict(cfg, g_ij_loss=True, g_kk_loss=False)
    train_metrics, intro, early = cmf_amd.get_non_square_train_metrics(cfg)
    opt = FlatOptimizer(dens.parameters(), opt="adam", lr=1e-3, max_grad_norm=100.0)
    x0 = g["x"][:8].float().cuda()
    losses = []
    from cmf_amd.training import train_batch
    for it in range(5):
        out = train_batch(dens, x0.clone(), 10_000, train_metrics, [opt])      # far past every warm-up boundary
        losses.append(float(out["metrics"]["loss"].detach()))
    ref_opt = torch.optim.Adam(dens.parameters(), lr=1e-3)                     # a torch optimiser drives the same path
    out = train_batch(dens, x0.clone(), 10_000, train_metrics, [ref_opt], max_grad_norm=100.0)
    losses.append(float(out["metrics"]["loss"].detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    opt.zero_grad()
    warm = train_metrics(dens, x0.clone(), 0)["loss"]                   # epoch 0: reconstruction-only objective
    warm.backward()
    assert float(opt.grad.abs().max()) > 0


def test_hutchinson_surrogate_gradients_match_oracle_autograd():
    """Train mode of ``hutch_with_cg`` (CIFAR configs): the surrogate mean_s u_s^T (J^T J eps_s) with u from CG, DETACHED as in
    the reference (the solve runs under no_grad, non_square.py:236-247), so its gradient is the Hutchinson estimate of
    d logdet / d theta.  With CG run to convergence u equals the exact solve: parameter gradients against autograd through
    the float64 oracle with u = solve(J^T J, eps).detach(); then the whole elbo through ``loss.backward()`` (finite, non-zero)."""
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build("mini_mnist")
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    named = dict(dens.named_parameters())
    B, d, S = 3, head.program.d, 3
    head.log_jacobian_method, head.num_hutchinson_samples, head.max_cg_iterations, head.cg_tolerance = "hutch_with_cg", S, 4 * d, 1e-7
    gen = torch.Generator().manual_seed(77)
    z_low, eps, a = g["z_low"][:B].float(), torch.randn(B, d, S, generator=gen), torch.randn(B, generator=gen)
    keys = [k for k, v in sd.items() if v.is_floating_point() and k in named]
    sd64 = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in sd.items()}
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    jtj, xh, J = O.jtj_batched(sd64, flow_ops, base, z_low.double())
    w = torch.bmm(jtj, eps.double())
    u = torch.linalg.solve(jtj, eps.double()).detach()
    value = (u * w).sum(1).mean(1)
    want = torch.autograd.grad((a.double() * value).sum(), [sd64[k] for k in keys], allow_unused=True)
    st = head.head_terms_forward(z_low.cuda(), tangents=True, hutch_eps=eps.cuda())
    assert rel(st["hutch"]["value"], value) < 1e-4 and rel(st["hutch"]["u"], u) < 1e-3
    out = head.head_terms_backward(z_low.cuda(), None, g_logdet=a.cuda(), state=st)
    checked = 0
    for k, wv in zip(keys, want):
        if wv is not None and float(wv.abs().max()) > 0:
            assert rel(out["grads"][named[k]], wv.reshape(named[k].shape)) < 2e-3, k
            checked += 1
    assert checked >= 40
    dens.train()
    dens.zero_grad()
    with torch.enable_grad():
        loss = -inner(dens, True).elbo(g["x"][:B].float().cuda(), add_reconstruction=True)["elbo"].mean()
        loss.backward()
    gn = torch.stack([p.grad.norm() for p in dens.parameters() if p.grad is not None])
    assert torch.isfinite(gn).all() and float(gn.max()) > 0


def test_training_memory_guard():
    """A training batch whose saved tangents cannot fit raises a clear error before anything is launched."""
    g, meta, cfg, dens = build("mini_mnist")
    head = find_head(dens)
    per = head.program.train_bytes_per_sample(16)
    assert per > 0
    B = int(400 * 2**30 // per) + 16                                   # > 288 GB worth of saved tangents
    x = torch.empty(B, 1, 1, 1, device="cuda").expand(B, *g["x"].shape[1:])   # no real storage: the guard fires first
    head.recompute = False                                             # (None would switch to recomputation per coupling layer)
    with pytest.raises(RuntimeError, match="smaller per-GPU batch"):
        head.train_forward(x)
    head.recompute = None
    assert head.program.train_bytes_per_sample(16, recompute=True) < per / 3


@pytest.mark.parametrize("name", ["mini_mnist", "c3_mnist_full"])
def test_recomputation_per_coupling_layer_gives_the_same_gradients(name):
    """``head.recompute = True``: each coupling layer keeps only its inputs and rebuilds its tangent state in the backward pass
    (the same kernels on the same data): identical loss, gradients equal to rounding."""
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    x = g["head_input"][:2].float().cuda() if "head_input" in g else g["x"][:2].float().cuda()
    kw = dict(add_offdiagonal_metric_reg=True)
    head.recompute = False
    loss_a, elbo_a, grads_a = head.loss_and_gradients(x.clone(), **kw)
    head.recompute = True
    loss_b, elbo_b, grads_b = head.loss_and_gradients(x.clone(), **kw)
    head.recompute = None
    assert torch.equal(elbo_a, elbo_b) and set(grads_a) == set(grads_b)
    for p in grads_a:
        assert rel(grads_b[p], grads_a[p]) < 1e-6


@pytest.mark.parametrize("name", ["c3_mnist_full"])           # 64-channel couplers: the small fixtures' networks do not qualify
def test_training_from_grouped_activations_and_bit_masks_gives_the_same_gradients(name, monkeypatch):
    """Batches of a multiple of 32 samples train from ``engine.ActList``: relu' bit masks written by the primal pass and the
    sample-grouped float activations, instead of 17 per-sample float activations per coupler regrouped forth and back.  Same
    kernels' arithmetic on the same data, so: identical loss, gradients equal to rounding (the weight gradients read the masks
    as bits instead of as floats, the forward tangent pass runs the bit-mask variant of the split kernel)."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    x0 = g["head_input"].float() if "head_input" in g else g["x"].float()
    gen = torch.Generator().manual_seed(5)
    x = x0[torch.randint(0, x0.shape[0], (32,), generator=gen)].cuda()
    x = x + 0.01 * torch.randn(x.shape, generator=gen).cuda()            # 32 distinct samples around the fixture's inputs
    kw = dict(add_offdiagonal_metric_reg=True)
    modes = []
    real = E.train_acts_mode
    monkeypatch.setattr(E, "train_acts_mode", lambda *a, **k: modes.append(real(*a, **k)) or modes[-1])
    loss_a, elbo_a, grads_a = head.loss_and_gradients(x.clone(), **kw)
    assert "train" in modes, "the batch of 32 did not take the ActList path"
    head.recompute = True                                # ... and with recomputation per coupling layer on the same path
    loss_r, elbo_r, grads_r = head.loss_and_gradients(x.clone(), **kw)
    head.recompute = None
    assert torch.equal(elbo_a, elbo_r)
    for p in grads_a:
        assert rel(grads_r[p], grads_a[p]) < 1e-6
    monkeypatch.setattr(E, "train_acts_mode", lambda *a, **k: True)
    loss_b, elbo_b, grads_b = head.loss_and_gradients(x.clone(), **kw)
    assert rel(elbo_a, elbo_b) < 1e-6 and set(grads_a) == set(grads_b)
    for p in grads_a:
        assert rel(grads_a[p], grads_b[p]) < 2e-6


def test_encode_layers_backward_full_size():
    """Every coupling layer of the full-size MNIST model's ENCODE chain, on the float64 oracle's own layer inputs (so no relu
    mask can differ): ``encode_train_`` / ``encode_backward_`` -- coupling backward, ScaledTanh stage, primal backward of the
    64-channel ResNet on the tangent-conv kernels -- against torch.autograd through the oracle's ``acl_x_to_z``."""
    from oracle import cmf_oracle as O
    from cmf_amd.bijections import AffineCouplingBijection
    g, meta, cfg, dens = build("c3_mnist_full")
    _, schema, x_shape, ops, sd = golden_model(meta, dtype=torch.float64)
    head = find_head(dens)
    pmap = dict(dens.named_parameters())
    pre, hd, flow_ops, base, prior_ops = O.split_ops(ops)
    x = g["x"][:2].double()
    h, _ = O.prehead(pre, x, torch.zeros_like(x))
    gen = torch.Generator().manual_seed(9)
    acls = iter([m for m in head.program.layers if isinstance(m, AffineCouplingBijection)])
    for op in flow_ops:
        k = op["kind"]
        if k == "flatten":
            h = h.flatten(1)
        elif k == "squeeze":
            h = O.squeeze_x_to_z(h, op["factor"])
        elif k == "split":
            h = torch.chunk(h, 2, dim=1)[0]
        if k != "acl":
            continue
        bij = next(acls)
        keys = [kk for kk in sd if kk.startswith(op["prefix"]) and sd[kk].is_floating_point() and kk in pmap]
        sd64 = {kk: (v.clone().requires_grad_(True) if kk in keys else v) for kk, v in sd.items()}
        xin = h.detach().clone().requires_grad_(True)
        z, _ = O.acl_x_to_z(sd64, op, xin)
        dz = torch.randn(z.shape, generator=gen).double() * (torch.rand(z.shape, generator=gen) < 0.3).double()
        want = torch.autograd.grad((z * dz).sum(), [sd64[kk] for kk in keys] + [xin], allow_unused=True)
        with torch.no_grad():
            hg = h.detach().float().cuda().contiguous()
            ctx = bij.encode_train_(hg)
            assert rel(hg, z) < 1e-6
            d, grads = dz.float().cuda().contiguous(), {}
            bij.encode_backward_(d, ctx, grads)
        assert rel(d, want[-1]) < 2e-6, op["prefix"]
        for kk, wv in zip(keys, want[:-1]):
            if wv is not None and float(wv.abs().max()) > 0:
                assert rel(grads[pmap[kk]], wv.reshape(pmap[kk].shape)) < 3e-5, kk     # hidden weight gradients: split-precision kernel
        h = z.detach()
