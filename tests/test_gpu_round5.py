"""Round-5 GPU tests: structural zeros in the decode sweep (the coupler fed by ``SplitDensity.pad_inputs``' zero channels runs no
tangent network and evaluates its primal network once), against the full computation -- bit for bit -- and the fixtures.
Every call goes through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_parity import build, find_head, rel

pytestmark = pytest.mark.gpu


def _batch(g, n, seed=11):
    """n distinct head inputs around the fixture's (the fixtures hold 2 - 3 samples)."""
    x0 = g["head_input"].float() if "head_input" in g else g["x"].float()
    gen = torch.Generator().manual_seed(seed)
    x = x0[torch.randint(0, x0.shape[0], (n,), generator=gen)]
    return (x + 0.01 * torch.randn(x.shape, generator=gen)).cuda()


class _full:
    """``with _full(prog):`` -- every coupling layer runs its whole network (rounds 1 - 4)."""

    def __init__(self, prog):
        self.prog = prog

    def __enter__(self):
        self.prog.SKIP_STRUCTURAL_ZEROS = False

    def __exit__(self, *exc):
        del self.prog.SKIP_STRUCTURAL_ZEROS
        return False


@pytest.mark.parametrize("name,B", [("mini_mnist", 3), ("mini_mnist", 32), ("mini_cifar", 3), ("mini_cifar", 16), ("c3_mnist_full", 2),
                                    ("c3_mnist_full", 32), ("c5_cifar_full", 16)])
def test_structural_zero_coupler_is_skipped_bit_for_bit(name, B):
    """x_hat, J, J^T J, log-det, g_ij, elbo and J^T w of the skipping decode sweep == the full one's, ``torch.equal``; the tangent
    network of exactly one coupler does not run (16 hidden launches fewer for the 8-block networks)."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    prog = head.program
    assert len(prog.zero_in) == 1
    x = _batch(g, B)
    out = {}
    with torch.no_grad():
        z_low = prog.encode(x)[0]
        Wd = torch.randn(B, int(np.prod(head.x_shape)), 3, device="cuda")
        for mode in ("skip", "full"):
            ctx = _full(prog) if mode == "full" else torch.no_grad()
            with ctx, E.timing(lambda n: n.startswith("conv_tangent")) as timer:
                x_hat, T = prog.decode(z_low, tangents=True)
                gr = E.gram_cholesky(T, prog.d)
                elbo = head.elbo(x.clone(), add_offdiagonal_metric_reg=True)["elbo"]
                xv, jw = prog.vjp(z_low, Wd)
                x_dec, _ = prog.decode(z_low, tangents=False)
                launches = {k: v[0] for k, v in timer.by_name().items()}
            out[mode] = dict(x_hat=x_hat.clone(), J=T.to_dense(prog.d).clone(), jtj=gr.jtj.clone(), logdet=gr.logdet.clone(),
                             l1=gr.l1_off.clone(), elbo=elbo.clone(), jw=jw.clone(), xv=xv.clone(), x_dec=x_dec.clone(), launches=launches)
    a, b = out["skip"], out["full"]
    for k in ("x_hat", "J", "jtj", "logdet", "l1", "elbo", "jw", "xv", "x_dec"):
        assert torch.equal(a[k], b[k]), (name, B, k, rel(a[k], b[k]))
    # the skipped coupler's hidden tangent convs: 2 per residual block, in each of the two tangent sweeps (decode, elbo) and in the
    # reverse sweep (vjp)
    net = prog.layers[next(iter(prog.zero_in))].net
    blocks = sum(1 for m in net.module if hasattr(m, "conv1"))
    hid = net.module[0].out_channels
    key = f"conv_tangent_t9_ci{hid}_co{hid}"
    assert b["launches"][key] - a["launches"][key] == 3 * (2 * blocks), (a["launches"], b["launches"])


@pytest.mark.parametrize("name,B,kw", [
    ("mini_mnist", 3, dict(add_offdiagonal_metric_reg=True)),
    ("mini_mnist", 32, dict(add_diagonal_metric_reg=True)),
    ("mini_cifar", 16, dict(add_offdiagonal_metric_reg=True)),
    ("c3_mnist_full", 32, dict(add_offdiagonal_metric_reg=True)),
])
def test_structural_zero_skip_gives_the_same_gradients(name, B, kw):
    """Training step with the skip == without: identical elbo, and every gradient tensor identical (``torch.equal``) where the
    full computation itself repeats bit for bit, else within twice its own run-to-run spread (atomic accumulation order).  Also
    with recomputation per coupling layer."""
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    prog = head.program
    x = _batch(g, B)
    with _full(prog):
        loss_f, elbo_f, grads_f = head.loss_and_gradients(x.clone(), **kw)
        loss_f2, elbo_f2, grads_f2 = head.loss_and_gradients(x.clone(), **kw)
    loss_s, elbo_s, grads_s = head.loss_and_gradients(x.clone(), **kw)
    head.recompute = True
    loss_r, elbo_r, grads_r = head.loss_and_gradients(x.clone(), **kw)
    head.recompute = None
    assert torch.equal(elbo_s, elbo_f) and torch.equal(elbo_r, elbo_f) and set(grads_s) == set(grads_f) == set(grads_r)
    exact = 0
    for p in grads_f:
        spread = rel(grads_f2[p], grads_f[p])
        if spread == 0.0:
            assert torch.equal(grads_s[p], grads_f[p]), (name, tuple(p.shape), rel(grads_s[p], grads_f[p]))
            exact += 1
        else:
            assert rel(grads_s[p], grads_f[p]) <= 2 * spread, (name, tuple(p.shape), rel(grads_s[p], grads_f[p]), spread)
        assert rel(grads_r[p], grads_f[p]) < 1e-6
    assert exact > 0


def test_structural_zero_skip_in_the_lowrank_hutchinson_training_step():
    """Train-mode ``hutch_with_cg`` on the full-size CIFAR model (S = 4: the low-rank backward, two tangent sweeps over one primal
    decode): same surrogate value and gradients with and without the skip."""
    g, meta, cfg, dens = build("c5_cifar_full")
    head = find_head(dens)
    prog = head.program
    head.log_jacobian_method, head.num_hutchinson_samples = "hutch_with_cg", 4
    head.train()
    x = _batch(g, 32)
    outs = []
    for mode in ("full", "skip"):
        torch.manual_seed(0)                                                # the same probes
        if mode == "full":
            with _full(prog):
                outs.append(head.loss_and_gradients(x.clone()))
        else:
            outs.append(head.loss_and_gradients(x.clone()))
        assert head.last_hutchinson.get("lowrank") is not None
    (lf, ef, gf), (ls, es, gs) = outs
    assert torch.equal(ef, es) and set(gf) == set(gs)
    for p in gf:
        assert rel(gs[p], gf[p]) < 1e-6, (tuple(p.shape), rel(gs[p], gf[p]))


def test_coupling_kernels_with_a_null_network_tangent_and_a_shared_network_output():
    """cmf_acl_tangent / cmf_acl_cotangent / cmf_acl_cross_terms with yt = NULL == the same call on an all-zero yt; cmf_acl_primal /
    cmf_acl_tangent with y_b = 0 == the same call on the row repeated B times."""
    from cmf_amd import engine as E
    from cmf_amd.bijections import SplitChannelwiseAffineCouplingBijection
    from cmf_amd.densities import FlowProgram  # noqa: F401
    gen = torch.Generator().manual_seed(2)
    B, C, H, W, nc = 5, 4, 6, 6, 16
    N = C * H * W

    class _Coupler(torch.nn.Module):
        shift_log_scale_net = None
    acl = SplitChannelwiseAffineCouplingBijection((C, H, W), lambda n: _Coupler(), reverse_mask=True)
    maps = acl.maps("cuda")
    z = torch.randn(B, C, H, W, generator=gen).cuda()
    y1 = (0.3 * torch.randn(1, 2 * acl.cmod, H, W, generator=gen)).cuda()
    g1 = torch.rand(1, 2 * acl.cmod, H, W, generator=gen).cuda()
    yB, gB = y1.expand(B, -1, -1, -1).contiguous(), g1.expand(B, -1, -1, -1).contiguous()
    T0 = torch.randn(B * N * nc, generator=gen).cuda()
    mk = lambda: E.Tangent(B, N, nc, "panel", "cuda", data=T0.clone())
    YT0 = E.Tangent(B, yB[0].numel(), nc, "panel", "cuda")
    YT0.data.zero_()
    # tangent update
    Ta, Tb, Tc = mk(), mk(), mk()
    E.acl_tangent(Ta, YT0, z, yB, gB, maps)
    E.acl_tangent(Tb, None, z, yB, gB, maps)
    E.acl_tangent(Tc, None, z, y1, g1, maps)
    assert torch.equal(Ta.data, Tb.data) and torch.equal(Ta.data, Tc.data)
    # primal update with a shared row
    za, zb = z.clone(), z.clone()
    E.acl_primal(za, yB, maps, decode=True)
    E.acl_primal(zb, y1, maps, decode=True)
    assert torch.equal(za, zb)
    lja, ljb = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    za, zb = z.clone(), z.clone()
    E.acl_primal(za, yB, maps, decode=True, lj=lja)
    E.acl_primal(zb, y1, maps, decode=True, lj=ljb)
    assert torch.equal(za, zb) and torch.equal(lja, ljb)
    # adjoint
    Ca, Cb = mk(), mk()
    YC = E.Tangent(B, yB[0].numel(), nc, "panel", "cuda")
    YC.data.zero_()
    E.acl_cotangent(Ca, YC, z, yB, gB, maps)
    E.acl_cotangent(Cb, None, z, y1, g1, maps)
    assert torch.equal(Ca.data, Cb.data)
    # cross terms
    V = E.modified_rows(mk(), maps)
    dza, dya, dga = torch.zeros_like(z), torch.zeros_like(yB), torch.zeros_like(gB)
    dyb, dgb = torch.zeros_like(yB), torch.zeros_like(gB)
    Ct = mk()
    E.acl_cross_terms(Ct, V, YT0, z, yB, gB, maps, dza, dya, dga)
    E.acl_cross_terms(Ct, V, None, z, yB, gB, maps, None, dyb, dgb)
    assert torch.equal(dya, dyb) and float(dza.abs().max()) == 0.0 and float(dga.abs().max()) == 0.0 and float(dgb.abs().max()) == 0.0


@pytest.mark.parametrize("hidden,B", [(128, 16), (128, 3), (192, 16)])
def test_wide_resnet_couplers_under_the_default_kernel_config(hidden, B):
    """g_hidden_channels of 128 / 192: the fp16-split primal kernel fetches ONE 64-channel group's bias per launch and
    refuses wider layers, so the sample-grouped primal pass must fall back to the fp32 kernel there -- elbo against the oracle,
    eval and one training step's gradients against the oracle's autograd."""
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    from oracle import cmf_oracle as O
    cfg = cmf_amd.get_config("mnist", g_hidden_channels=[hidden], latent_dimension=8, log_jacobian_method="cholesky")
    schema = cmf_amd.get_schema(cfg)
    gen = torch.Generator().manual_seed(hidden + B)
    x = torch.randint(0, 256, (B, 1, 28, 28), generator=gen).float() + torch.rand(B, 1, 28, 28, generator=gen)
    density = cmf_amd.get_density(schema, x)
    sd = fill_state_dict(density.state_dict(), seed=0)
    density.load_state_dict(sd)
    density = density.cuda().eval()
    inner = density.module.density
    with torch.no_grad():
        got = inner.elbo(x.cuda(), add_reconstruction=True, add_offdiagonal_metric_reg=True)["elbo"]
        ops = O.compile_schema(schema, (1, 28, 28))
        want = O.elbo(sd, ops, x, add_offdiagonal_metric_reg=True, noise=torch.zeros_like(x))["elbo"]
    assert rel(got, want) < 1e-4
    head = find_head(density)
    pre, _, flow_ops, base, prior_ops = O.split_ops(ops)
    with torch.no_grad():
        y, pre_lj = O.prehead(pre, x, noise=torch.zeros_like(x))
    loss, elbo, grads = head.loss_and_gradients(y.cuda(), add_offdiagonal_metric_reg=True)
    assert rel(elbo, want - pre_lj.view(-1, 1)) < 1e-4
    assert all(torch.isfinite(v).all() for v in grads.values())


@pytest.mark.parametrize("H,W", [(14, 14), (28, 28), (16, 16), (32, 32), (4, 14), (8, 8)])
@pytest.mark.parametrize("live", [1, 2])
@pytest.mark.parametrize("fmode", ["relu", "bits"])
@pytest.mark.parametrize("with_res", [True, False])
def test_checkerboard_output_of_the_split_kernel(H, W, live, fmode, with_res):
    """cmf_conv_tangent_bf16x3 with ``live`` = 1 / 2 (the last hidden conv of a checkerboard coupler's network, acl.py:48-66): the
    compact output == the full launch's at the pixels with (row + col) % 2 == live - 1, bit for bit, and == F.conv2d x mask in
    float64 to the split arithmetic's accuracy; the residual is read at those pixels of the full image."""
    import torch.nn.functional as F
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(H * 100 + W + live)
    B, C, nc, HW = 3, 64, 32, H * W
    S = nc // 16
    x = torch.randn(B, C, H, W, nc, generator=gen)
    prim = torch.randn(B, C, H, W, generator=gen)
    res = torch.randn(B, C, H, W, nc, generator=gen)
    w = torch.randn(C, C, 3, 3, generator=gen) / 24
    to_dev = lambda t: t.reshape(B, C, -1, S, 16).permute(0, 2, 3, 1, 4).contiguous().cuda()     # slice-major [px][slice][ch][16]
    st, sl = (C * HW * nc, 16, C * nc), C * 16
    wd = torch.nn.Parameter(w.cuda())
    xd, rd = to_dev(x), to_dev(res) if with_res else None
    bits = E.relu_bits(prim.cuda())
    fk = dict(fmode=E.F_RELU_BITS, f=bits.data, f_np=bits.np_bytes) if fmode == "bits" else dict(fmode=E.F_RELU, f=prim.cuda(), f_np=C * HW, f_ci=HW, f_px=1)
    full = torch.empty(B, HW, S, C, 16, device="cuda")
    E.conv_tangent(xd, 0, *st, wd, 9, full, *st, B, C, C, H, W, nc, res_t=rd, x_sl=sl, y_sl=sl, precision="bf16x3", **fk)
    comp = torch.full((B, HW // 2, S, C, 16), float("nan"), device="cuda")
    E.conv_tangent(xd, 0, *st, wd, 9, comp, st[0] // 2, st[1], st[2], B, C, C, H, W, nc, res_t=rd, x_sl=sl, y_sl=sl, precision="bf16x3",
                   live=live, res_np=st[0], **fk)
    ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    sel = ((ii + jj) % 2 == live - 1).reshape(-1)                       # row-major order of the live pixels = the compact index
    assert int(sel.sum()) == HW // 2
    assert torch.equal(comp, full[:, sel.cuda()])
    want = F.conv2d((x * (prim > 0).float().unsqueeze(-1)).permute(0, 4, 1, 2, 3).reshape(B * nc, C, H, W).double(), w.double(), padding=1)
    want = want.reshape(B, nc, C, HW).permute(0, 3, 1, 2).reshape(B, HW, S, 16, C).permute(0, 1, 2, 4, 3)
    if with_res:
        want = want + to_dev(res).cpu().double()
    assert rel(comp, want[:, sel]) < 2e-5


@pytest.mark.parametrize("name,B", [("c3_mnist_full", 2), ("c3_mnist_full", 32), ("c5_cifar_full", 16), ("c3_mnist_full_cond", 2)])
def test_checkerboard_tail_of_the_coupler_networks_is_bit_identical(name, B, monkeypatch):
    """The seven checkerboard couplers' last hidden conv + 1x1 conv at the (1 - mask) pixels only (engine.net_tangent, compact
    output) against the full-image form: x_hat, J, J^T J, log-det, g_ij and elbo ``torch.equal``; 7 launches take the
    checkerboard form per tangent sweep."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    prog = head.program
    x = _batch(g, B)
    out = {}
    with torch.no_grad():
        z_low = prog.encode(x)[0]
        for mode in (True, False):
            monkeypatch.setattr(E, "CHECKERBOARD_TAIL", mode)
            with E.timing(lambda n: n.startswith("conv_tangent")) as timer:
                x_hat, T = prog.decode(z_low, tangents=True)
                gr = E.gram_cholesky(T, prog.d)
                launches = {k: v[0] for k, v in timer.by_name().items()}
            elbo = head.elbo(x.clone(), add_offdiagonal_metric_reg=True)["elbo"]
            out[mode] = dict(x_hat=x_hat.clone(), J=T.to_dense(prog.d).clone(), jtj=gr.jtj.clone(), logdet=gr.logdet.clone(), l1=gr.l1_off.clone(),
                             elbo=elbo.clone(), launches=launches)
    for k in ("x_hat", "J", "jtj", "logdet", "l1", "elbo"):
        assert torch.equal(out[True][k], out[False][k]), (name, B, k, rel(out[True][k], out[False][k]))
    assert out[True]["launches"].get("conv_tangent_t9_ci64_co64_live") == 7 and "conv_tangent_t9_ci64_co64_live" not in out[False]["launches"]
    assert out[False]["launches"]["conv_tangent_t9_ci64_co64"] - out[True]["launches"]["conv_tangent_t9_ci64_co64"] == 7


def test_bench_launches_four_ranks_from_a_bare_shell_and_tears_down():
    """Rehearsal of the driver's multi-GPU launch on a one-GPU box: ``python bench.py --gpus 4`` from a bare environment (fresh
    children, never a re-exec; gloo + one shared GPU), every N > 1 leg of the line present and seen by 4 ranks, and the job gone
    within 30 s of rank 0 printing its line (process-group destruction and exit of every rank).  FOUR ranks, not eight: this pool
    allows at most 6 processes on a card at once and the test process itself holds the GPU (4 + 1 <= 6); the 8-rank launch path
    is the same code with a larger range()."""
    import json, os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--backend", "gloo", "--share-gpu", "--batch", "8",
                        "--steps", "1", "--warmup", "1", "--leg-steps", "1", "--train-batch", "8", "--cpu-batch", "0"],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=root)
    t_exit = time.time()
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["config"]["global_batch"] == 32 and "REHEARSAL" in d["data"]
    for leg in ("strong_c3", "c5", "grad_reduce", "train"):
        assert leg in d and "error" not in d[leg], leg
    assert d["strong_c3"]["ranks_seen"] == 4 and d["strong_c3"]["per_gpu_batch"] == 128 and d["strong_c3"]["global_batch"] == 512
    assert d["c5"]["ranks_seen"] == 4 and d["c5"]["global_batch"] == 128
    assert d["train"]["n_gpus"] == 4 and d["train"]["config"]["global_batch"] == 32 and d["train"]["config"]["gradient_reduction"] == d["grad_reduce"]["used"]
    assert 0 <= t_exit - d["printed_at_unix"] < 30, (t_exit - d["printed_at_unix"], t_exit - t0)


@pytest.mark.parametrize("name,B", [("c3_mnist_full", 2), ("c3_mnist_full", 32), ("c5_cifar_full", 16), ("c3_mnist_full_cond", 2)])
def test_first_coupler_on_its_live_seed_columns_is_bit_identical(name, B):
    """The first decoded coupler's tangent network on the Jacobian columns seeded at its pass-through elements only (packed,
    expanded back by cmf_expand_columns) against all d columns: x_hat, J, J^T J, log-det, g_ij, elbo ``torch.equal``; and the
    column expansion itself against torch indexing."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    prog = head.program
    x = _batch(g, B)
    plan = prog._seed_columns(x.device)
    assert plan is not None and plan["nc"] < E.ceil16(prog.d) and plan["n"] <= plan["nc"]
    out = {}
    with torch.no_grad():
        z_low = prog.encode(x)[0]
        for mode in (True, False):
            prog.SEED_COLUMNS = mode
            try:
                x_hat, T = prog.decode(z_low, tangents=True)
                gr = E.gram_cholesky(T, prog.d)
                elbo = head.elbo(x.clone(), add_offdiagonal_metric_reg=True)["elbo"]
            finally:
                del prog.SEED_COLUMNS
            out[mode] = dict(x_hat=x_hat.clone(), J=T.to_dense(prog.d).clone(), jtj=gr.jtj.clone(), logdet=gr.logdet.clone(), l1=gr.l1_off.clone(),
                             elbo=elbo.clone())
    for k in ("x_hat", "J", "jtj", "logdet", "l1", "elbo"):
        assert torch.equal(out[True][k], out[False][k]), (name, B, k, rel(out[True][k], out[False][k]))
    # the expansion kernel
    gen = torch.Generator().manual_seed(1)
    src = E.Tangent(3, 50, 32, "panel", "cuda", data=torch.randn(3 * 50 * 32, generator=gen).cuda())
    cmap = torch.full((64,), -1, dtype=torch.int32)
    sel = torch.randperm(64, generator=gen)[:20]
    cmap[sel] = torch.randperm(32, generator=gen)[:20].to(torch.int32)
    got = E.expand_columns(src, 64, cmap.cuda()).data.view(150, 64).cpu()
    want = torch.zeros(150, 64)
    want[:, sel] = src.data.view(150, 32).cpu()[:, cmap[sel].long()]
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,H,W,nc", [(1, 2, 14, 16), (2, 28, 14, 128), (5, 4, 8, 16), (1, 12, 32, 64), (2, 14, 28, 48), (7, 2, 28, 16), (1, 8, 16, 80)])
def test_checkerboard_output_odd_sizes(B, H, W, nc):
    """More shapes of the checkerboard-output form: one tile, one sample, 1 / 3 / 5 / 8 column slices, non-square images, both tile
    families; both parities, relu' from bits, with residual: compact == full launch at the live pixels, bit for bit."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(B * 1000 + H * 10 + W + nc)
    C, HW, S = 64, H * W, nc // 16
    x = torch.randn(B, HW, S, C, 16, generator=gen).cuda()                      # slice-major [px][slice][ch][16]
    res = torch.randn(B, HW, S, C, 16, generator=gen).cuda()
    prim = torch.randn(B, C, H, W, generator=gen).cuda()
    wd = torch.nn.Parameter((torch.randn(C, C, 3, 3, generator=gen) / 24).cuda())
    st, sl = (C * HW * nc, 16, C * nc), C * 16
    bits = E.relu_bits(prim)
    fk = dict(fmode=E.F_RELU_BITS, f=bits.data, f_np=bits.np_bytes)
    full = torch.empty(B, HW, S, C, 16, device="cuda")
    E.conv_tangent(x, 0, *st, wd, 9, full, *st, B, C, C, H, W, nc, res_t=res, x_sl=sl, y_sl=sl, precision="bf16x3", **fk)
    ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for live in (1, 2):
        comp = torch.full((B, HW // 2, S, C, 16), float("nan"), device="cuda")
        E.conv_tangent(x, 0, *st, wd, 9, comp, st[0] // 2, st[1], st[2], B, C, C, H, W, nc, res_t=res, x_sl=sl, y_sl=sl, precision="bf16x3",
                       live=live, res_np=st[0], **fk)
        sel = ((ii + jj) % 2 == live - 1).reshape(-1).cuda()
        assert torch.equal(comp, full[:, sel]), (B, H, W, nc, live)


def test_seed_column_plan_follows_the_permutation_buffer():
    """A new permutation written into the tail's buffer (``load_state_dict`` does exactly that) gives a new plan -- the cache is keyed on
    the buffer's version -- and the packed form stays bit-identical to the full one for every draw, incl. one with NO live column."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build("c3_mnist_full")
    head = find_head(dens)
    prog = head.program
    x = _batch(g, 16)
    tail, first = prog.tail, prog.layers[-1]
    gen = torch.Generator().manual_seed(9)
    seen = set()
    mod = torch.from_numpy(first._maps._host["zi"]).long()                      # modified elements of the first decoded coupler
    for draw in range(4):
        perm = torch.randperm(tail.flattened_dims, generator=gen)
        if draw == 3:                                                            # every latent on a MODIFIED element: no live column
            rest = torch.tensor(sorted(set(range(tail.flattened_dims)) - set(mod[: prog.d].tolist())))
            perm = torch.cat((mod[: prog.d], rest))
        with torch.no_grad():
            tail.permutation.copy_(perm.cuda())
            tail.inverse_permutation.copy_(torch.argsort(perm).cuda())
        plan = prog._seed_columns(x.device)
        seen.add(None if plan is None else (plan["n"], plan["nc"]))
        if draw == 3:
            assert plan is not None and plan["n"] == 0 and plan["nc"] == 16
        outs = []
        with torch.no_grad():
            z_low = prog.encode(x)[0]
            for mode in (True, False):
                prog.SEED_COLUMNS = mode
                try:
                    x_hat, T = prog.decode(z_low, tangents=True)
                finally:
                    del prog.SEED_COLUMNS
                outs.append((x_hat.clone(), T.to_dense(prog.d).clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), draw
    assert len(seen) >= 2


@pytest.mark.parametrize("tangent,primal", [("f32", "f32"), ("f32", "f16x3"), ("bf16x3", "f32")])
@pytest.mark.parametrize("name,B", [("c3_mnist_full", 16), ("c5_cifar_full", 3)])
def test_all_skipping_forms_under_the_other_kernel_configs(name, B, tangent, primal, monkeypatch):
    """Structural zeros + checkerboard tail + seed columns together against none of them, under the exact-fp32 tangent kernel and
    the exact-fp32 primal kernel (``head.kernels``): x_hat, J, J^T J, elbo ``torch.equal``."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    head.kernels = E.KernelConfig(tangent=tangent, primal=primal)
    prog = head.program
    x = _batch(g, B)
    outs = []
    with torch.no_grad():
        z_low = prog.encode(x)[0]
        for on in (True, False):
            monkeypatch.setattr(E, "CHECKERBOARD_TAIL", on)
            prog.SKIP_STRUCTURAL_ZEROS = prog.SEED_COLUMNS = on
            try:
                x_hat, T = prog.decode(z_low, tangents=True)
                gr = E.gram_cholesky(T, prog.d)
                elbo = head.elbo(x.clone(), add_offdiagonal_metric_reg=True)["elbo"]
            finally:
                del prog.SKIP_STRUCTURAL_ZEROS, prog.SEED_COLUMNS
            outs.append((x_hat.clone(), T.to_dense(prog.d).clone(), gr.jtj.clone(), elbo.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b), (name, tangent, primal, rel(a, b))


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar_cond1e3", "c3_mnist_full", "c5_cifar_full"])
def test_reference_vectors_through_the_full_forms_too(name, monkeypatch):
    """The fixture tests run through the skipping forms (the defaults); here the same reference vectors through the FULL forms
    (every switch off), so that the rounds 1 - 4 path stays pinned as well."""
    from cmf_amd import engine as E
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    prog = head.program
    monkeypatch.setattr(E, "CHECKERBOARD_TAIL", False)
    prog.SKIP_STRUCTURAL_ZEROS = prog.SEED_COLUMNS = False
    dequant = "noise" in g
    x = (g["x"] + g["noise"]) if dequant else g["x"]
    core = dens.module.density if dequant else dens
    with torch.no_grad():
        for i, (lw, mw, rec, off, diag) in enumerate(meta["elbo_combos"]):
            out = core.elbo(x.cuda(), likelihood_wt=lw, metric_wt=mw, add_reconstruction=rec, add_offdiagonal_metric_reg=off,
                            add_diagonal_metric_reg=diag)
            assert rel(out["elbo"], g[f"elbo_{i}"]) < 1e-4, (name, i)
        x_hat, T = prog.decode(g["z_low"].cuda(), tangents=True)
        gf = E.gram_cholesky(T, prog.d)
        assert rel(x_hat, g["x_hat"]) < 1e-5 and rel(gf.jtj, g["jtj"]) < 1e-4 and rel(gf.logdet.view(-1, 1), g["logdet"]) < 1e-4
