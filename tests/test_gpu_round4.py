"""Round-4 GPU tests: the fp16-split primal convolutions (cmf_conv_tangent_f16x3) -- values, relu' bit masks, the input-range
chain, relu-mask flips against the float64 oracle -- and the per-head kernel configuration.  Every call goes through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, golden_model
from test_gpu_parity import build, find_head, inner, rel

pytestmark = pytest.mark.gpu


def _unpack_bits(mask, C):
    """BitMask.data (B, HW, C/8) uint8 -> bool (B, HW, C): bit j of byte o = channel 8 o + j."""
    return ((mask.to(torch.int32).unsqueeze(-1) >> torch.arange(8, device=mask.device, dtype=torch.int32)) & 1).bool().flatten(2)


@pytest.mark.parametrize("H,W", [(14, 14), (28, 28), (16, 16), (8, 32)])
@pytest.mark.parametrize("scale", [1.0, 3e4, 1e-4])
def test_f16x3_primal_conv_values_masks_and_range_chain(H, W, scale):
    """One hidden primal conv with 16 samples in the column slots on the fp16-split kernel: relu on load, bias, residual against
    float64 ``F.conv2d`` (fp32-grade: the split is 11 + 11 significant bits); the sign bits it writes for the next tangent conv;
    the running maximum it hands to the next conv.  ``scale``: inputs far outside fp16's exponent range in either direction (the
    power-of-two input scale derived from amax_in keeps them exact)."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(H * W)
    B, C, HW = 32, 64, H * W
    G = B // 16
    x = scale * torch.randn(B, C, H, W, generator=gen)
    w = torch.randn(C, C, 3, 3, generator=gen) / 24
    bias = scale * torch.randn(C, generator=gen)
    res = scale * torch.randn(B, C, H, W, generator=gen)
    want = F.conv2d(torch.relu(x).double(), w.double(), bias.double(), padding=1) + res.double()
    xg, rg = E.primal_regroup(x.cuda(), True), E.primal_regroup(res.cuda(), True)
    wd = torch.nn.Parameter(w.cuda())
    pn = (C * HW * 16, HW * 16, 16)
    rng = torch.zeros(2, device="cuda")
    E.absmax(xg, rng[0:1])
    assert float(rng[0]) == float(x.abs().max())
    outs = {}
    for prec, item in (("f16x3", 64), ("f32", 0), ("f16x3", 32)):
        yg = torch.empty_like(xg)
        m = E.BitMask(B, HW, C, "cuda")
        m.data.fill_(0xAA)
        rng[1] = 0
        kw = dict(amax_in=rng[0:1], amax_out=rng[1:2], item_channels=item) if prec == "f16x3" else {}
        E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias.cuda(), res_t=rg, precision=prec,
                       mask_out=m.data, mask_np=m.np_bytes, **kw)
        got = E.primal_regroup(yg.view(G, -1), False).view(B, C, H, W)
        outs[prec, item] = (got, m, float(rng[1]))
    # 32-channel work items (launches with few items: cmf_conv_tangent_f16x3's own choice at these sizes) = the 64-channel ones, bit
    # for bit: values, sign bits, running maximum -- also without a residual (the other kernel variant)
    assert torch.equal(outs["f16x3", 32][0], outs["f16x3", 64][0]) and torch.equal(outs["f16x3", 32][1].data, outs["f16x3", 64][1].data)
    assert outs["f16x3", 32][2] == outs["f16x3", 64][2]
    plain = {}
    for item in (64, 32, 0):
        yg = torch.empty_like(xg)
        m = E.BitMask(B, HW, C, "cuda")
        m.data.fill_(0x55)
        rng[1] = 0
        E.conv_tangent(xg, 0, *pn, wd, 9, yg, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias.cuda(), precision="f16x3",
                       mask_out=m.data, mask_np=m.np_bytes, amax_in=rng[0:1], amax_out=rng[1:2], item_channels=item)
        plain[item] = (E.primal_regroup(yg.view(G, -1), False).view(B, C, H, W), m.data.clone(), float(rng[1]))
    for item in (32, 0):
        assert torch.equal(plain[item][0], plain[64][0]) and torch.equal(plain[item][1], plain[64][1]) and plain[item][2] == plain[64][2]
    assert rel(plain[64][0], want - res.double()) < 1e-6
    assert torch.equal(_unpack_bits(plain[64][1], C), (plain[64][0] > 0).permute(0, 2, 3, 1).reshape(B, HW, C))
    got, m, _ = outs["f16x3", 64]
    rng[1] = outs["f16x3", 64][2]
    e16, e32 = rel(got, want), rel(outs["f32", 0][0], want)
    # fp32-grade: within a small factor of the exact-fp32-product kernel (whose residual is added once, after the products; here it
    # is the accumulators' initial value, so every partial sum is rounded at the residual's magnitude -- like the bf16 split kernel)
    assert e16 < 1e-6 and e16 < 3 * e32 + 1e-7, (e16, e32)
    # the bit mask is the sign of what was stored, bit for bit
    want_bits = (got > 0).permute(0, 2, 3, 1).reshape(B, HW, C)
    assert torch.equal(_unpack_bits(m.data, C), want_bits)
    # the running maximum covers every positive stored value (what the next conv's relu-on-load lets through)
    assert float(rng[1]) == float(got.clamp_min(0).max())
    # without a range word the kernel still works for inputs inside fp16's range
    if scale == 1.0:
        y2 = torch.empty_like(xg)
        E.conv_tangent(xg, 0, *pn, wd, 9, y2, *pn, G, C, C, H, W, 16, fmode=E.F_SELF_RELU, bias=bias.cuda(), res_t=rg, precision="f16x3")
        assert rel(E.primal_regroup(y2.view(G, -1), False).view(B, C, H, W), want) < 1e-6


@pytest.mark.parametrize("H,W", [(14, 14), (28, 28), (16, 16), (8, 32)])
@pytest.mark.parametrize("scale", [1.0, 1e-5])
def test_f16x3_primal_backward_conv(H, W, scale):
    """The data-gradient conv of the primal backward on the fp16-split kernel (16 samples in the column slots): plain cotangent in,
    the ADJOINT operator packed from the layer's weight, per-sample relu' of the forward activation on the way out, optional
    residual -- against float64 autograd of ``conv2d(relu(a))`` and against the fp32 kernel's form of the same launch; both work-item
    sizes bit-identical; *amax_out = max |stored| (the next conv's input range).  ``scale``: cotangents far below fp16's range."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(H + W)
    B, C, HW = 32, 64, H * W
    G = B // 16
    act = torch.randn(B, C, H, W, generator=gen)                     # the forward activation whose relu the cotangent passes through
    w = torch.randn(C, C, 3, 3, generator=gen) / 24
    gy = scale * torch.randn(B, C, H, W, generator=gen)              # cotangent of y = conv2d(relu(act), w)
    res = scale * torch.randn(B, C, H, W, generator=gen)
    a64 = act.double().requires_grad_(True)
    F.conv2d(torch.relu(a64), w.double(), None, padding=1).backward(gy.double())
    want = a64.grad                                                   # = [act > 0] . conv_transpose(gy, w)
    gg, ag, rg = (E.primal_regroup(t.cuda(), True) for t in (gy, act, res))
    wd = torch.nn.Parameter(w.cuda())
    pn = (C * HW * 16, HW * 16, 16)
    fo = dict(fo=ag, fo_np=C * HW * 16, fo_co=HW * 16, fo_px=16, fomode=E.F_SELF_RELU)
    rng = torch.zeros(2, device="cuda")
    E.absmax(gg, rng[0:1])
    for with_res in (False, True):
        ref = want + res.double() if with_res else want
        outs = {}
        for prec, item in (("f32", 0), ("f16x3", 64), ("f16x3", 32), ("f16x3", 0)):
            yg = torch.full_like(gg, float("nan"))
            rng[1] = 0
            kw = dict(amax_in=rng[0:1], amax_out=rng[1:2], item_channels=item) if prec == "f16x3" else {}
            with E.timing(lambda n: True) as timer:
                E.conv_tangent(gg, 0, *pn, wd, 9, yg, *pn, G, C, C, H, W, 16, transpose=True, precision=prec, res_t=rg if with_res else None,
                               **fo, **kw)
            assert list(timer.by_name()) == ["conv_tangent_t9_ci64_co64_primal_bwd"]
            outs[prec, item] = (E.primal_regroup(yg.view(G, -1), False).view(B, C, H, W), float(rng[1]))
        e32 = rel(outs["f32", 0][0], ref)
        for item in (64, 32, 0):
            got, amax = outs["f16x3", item]
            assert torch.equal(got, outs["f16x3", 64][0]) and amax == outs["f16x3", 64][1]
            assert rel(got, ref) < 1e-6 and rel(got, ref) < 3 * e32 + 1e-7, (rel(got, ref), e32)
            assert amax == float(got.abs().max())
            if not with_res:
                assert bool((got[act.cuda() <= 0] == 0).all())                     # masked exactly, not approximately


def test_f16x3_pack_single_and_batched_agree():
    """cmf_pack_weight_f16x3 (two launches: scale, then pack) and kind 2 of cmf_pack_weights_batched write the same bytes, the
    trailer holds the power of two that puts max |w| in [2^11, 2^12), forward and adjoint operator."""
    import ctypes as C
    from cmf_amd import engine as E, _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(5)
    for cout, cin, tr, gain in ((64, 64, 0, 0.04), (64, 64, 1, 7.0), (128, 64, 0, 1e-3), (64, 32, 1, 300.0)):
        w = (gain * torch.randn(cout, cin, 3, 3, generator=gen)).cuda()
        pc, pi = (cin, cout) if tr else (cout, cin)                    # the packed operator's channels
        n = C.c_longlong(0)
        _lib.check(lib.cmf_pack_weight_f16x3(None, None, pc, pi, tr, C.byref(n), None), "size")
        one, two = torch.zeros(n.value, dtype=torch.uint8, device="cuda"), torch.ones(n.value, dtype=torch.uint8, device="cuda")
        _lib.check(lib.cmf_pack_weight_f16x3(E._p(w), E._p(one), pc, pi, tr, None, E._stream()), "pack")
        desc = np.zeros(1, dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("total", "<i8"), ("cout", "<i4"), ("cin", "<i4"), ("taps", "<i4"),
                                           ("transpose", "<i4"), ("kind", "<i4"), ("reserved", "<i4")]))
        desc[0] = (w.data_ptr(), two.data_ptr(), (n.value - 16) // 2, pc, pi, 9, tr, 2, 0)
        table = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
        _lib.check(lib.cmf_pack_weights_batched(E._p(table), 1, E._stream()), "batched")
        assert torch.equal(one, two)
        trailer = one[-16:].view(torch.float32).cpu()
        k = float(torch.log2(trailer[0]))
        assert k == round(k) and float(trailer[0] * trailer[1]) == 1.0 and 2 ** 11 <= float(w.abs().max()) * 2 ** k < 2 ** 12
        halves = one[:-16].view(torch.float16).float()
        assert bool(torch.isfinite(halves).all()) and float(halves.abs().max()) < 2 ** 12


def _oracle_hidden_chain(sd, op, x):
    """float64 pre-activations of every hidden layer of a ResNet coupler (oracle.net_forward's chain, networks.py:50-60)."""
    from oracle import cmf_oracle as O
    p = O._net_keys(op)
    n = len(op["hidden"])
    d = lambda k: sd[k].double()
    h = F.conv2d(x.double(), d(p + "module.0.weight"), None, padding=1)
    acts = [h]
    for i in range(1, n + 1):
        o = F.conv2d(torch.relu(h), d(f"{p}module.{i}.conv1.weight"), d(f"{p}module.{i}.conv1.bias"), padding=1)
        h = F.conv2d(torch.relu(o), d(f"{p}module.{i}.conv2.weight"), d(f"{p}module.{i}.conv2.bias"), padding=1) + h
        acts += [o, h]
    return acts


def test_relu_mask_flips_of_the_f16x3_primal_against_the_float64_oracle():
    """the relu masks the tangent pass takes from the primal activations.  Full-size MNIST model, the first 16
    inputs of ``c3_mnist_stats32`` decoded layer by layer by the ORACLE (float64); at every coupling layer the HIP primal pass runs
    on the oracle's layer input in both arithmetics and its bit masks are compared with the signs of the oracle's float64
    activations.  The fp16-split primal may flip at most 1.5 x the masks the exact-fp32-product primal flips (+ 3 sigma of a
    Poisson count: the counts are a dozen out of 10^8), and a small multiple of the flips the float32 oracle itself shows."""
    import cmf_amd
    from cmf_amd import engine as E
    from cmf_amd.bijections import AffineCouplingBijection
    from cmf_amd.recipe import fill_state_dict
    from oracle import cmf_oracle as O
    g, meta = load_golden("c3_mnist_stats32")
    _, _, _, ops, sd = golden_model(meta)
    cfg = cmf_amd.get_config("mnist", **meta["overrides"])
    dens = cmf_amd.get_density(cmf_amd.get_schema(cfg), torch.zeros(4, 1, 28, 28))
    dens.load_state_dict(fill_state_dict(dens.state_dict(), seed=meta["recipe_seed"]), strict=True)
    dens = dens.cuda().eval()
    prog = find_head(dens).program
    pre, head, flow_ops, base, prior_ops = O.split_ops(ops)
    n = 16
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    z = O.tail_scatter(sd64, base, g["z_low"][:n].double())
    layers = [m for m in prog.layers]
    assert len(layers) == len(flow_ops)
    flips, total, flips32cpu = {"f16x3": 0, "f32": 0}, 0, 0
    for m, op in zip(reversed(layers), reversed(flow_ops)):
        k = op["kind"]
        if k == "acl":
            assert isinstance(m, AffineCouplingBijection)
            # the network input as the oracle's ACL forms it (mask . z for the checkerboard layers, the pass-through channels else)
            view = m.view("cuda")
            zf = z.float().cuda()
            xin = sd64[op["prefix"] + "mask"] * z if op["mask_type"] == "checkerboard" else O._cw_split(op, z)[0]
            acts64 = _oracle_hidden_chain(sd64, op, xin)
            acts32 = _oracle_hidden_chain(sd, op, xin.float())       # the float32 oracle (ATen CPU convs) on the same input
            want = [(a > 0) for a in acts64[1:-1]]                    # the activations whose masks the tangent pass reads as bits
            flips32cpu += sum(int(((a32 > 0) != w).sum()) for a32, w in zip(acts32[1:-1], want))
            for prec in ("f16x3", "f32"):
                with E.scope(primal=prec):
                    y, gg, acts = E.net_primal(m.net, zf, view, need_acts="bits")
                masks = [a for a in acts if isinstance(a, E.BitMask)]
                assert len(masks) == len(want)
                for bm, w in zip(masks, want):
                    got = _unpack_bits(bm.data, w.shape[1]).reshape(n, view.geom.H, view.geom.W, -1).permute(0, 3, 1, 2).cpu()
                    flips[prec] += int((got != w).sum())
            total += sum(w.numel() for w in want)
            z = O.acl_z_to_x(sd64, op, z)
        elif k == "flatten":
            z = z.reshape(z.shape[0], *op["x_shape"])
        elif k == "squeeze":
            z = O.squeeze_z_to_x(z, op["factor"])
        elif k == "split":
            z = torch.cat((z, torch.zeros_like(z)), 1)
    print(f"relu-mask flips vs the float64 oracle over {total:.2e} mask bits: f16x3 primal {flips['f16x3']}, fp32-MFMA primal {flips['f32']}, "
          f"float32 CPU oracle {flips32cpu}")
    assert flips["f16x3"] <= 1.5 * flips["f32"] + 3 * max(flips["f32"], 1) ** 0.5 + 2
    assert flips["f16x3"] <= 3 * max(flips32cpu, 4)


def test_kernel_config_is_per_head_and_thread_local():
    """VERDICT r3 item 8 / SURVEY 8b: the kernels' arithmetic is an attribute of the head (``head.kernels``), carried by a
    thread-local scope.  Two threads evaluating two heads with different configurations at the same time get the results of the
    sequential runs bit for bit; nothing is process-global."""
    import copy
    import threading
    from cmf_amd import engine as E
    import cmf_amd
    from cmf_amd.recipe import fill_state_dict
    cfg = cmf_amd.get_config("mnist", latent_dimension=16, g_hidden_channels=[64], log_jacobian_method="cholesky")   # 64 hidden channels:
    dens_a = cmf_amd.get_density(cmf_amd.get_schema(cfg), torch.zeros(1, 1, 28, 28))                               # the split kernels run
    dens_a.load_state_dict(fill_state_dict(dens_a.state_dict(), seed=3), strict=True)
    dens_a = dens_a.cuda().eval()
    dens_b = copy.deepcopy(dens_a)
    ha, hb = find_head(dens_a), find_head(dens_b)
    ha.kernels = E.KernelConfig(tangent="bf16x3", primal="f16x3")
    hb.kernels = E.KernelConfig(tangent="f32", primal="f32")
    assert E.cfg().tangent == "bf16x3" and E.cfg().primal == "f16x3"           # the defaults, untouched by the heads
    gen = torch.Generator().manual_seed(9)
    x = (torch.randint(0, 256, (32, 1, 28, 28), generator=gen).float() + torch.rand(32, 1, 28, 28, generator=gen)).cuda()
    kw = dict(add_reconstruction=True, add_offdiagonal_metric_reg=True)

    def run(d):
        with torch.no_grad():
            return inner(d, True).elbo(x.clone(), **kw)["elbo"].clone()

    want = [run(dens_a), run(dens_b)]
    torch.cuda.synchronize()
    got, errs, seen = [None, None], [], [None, None]

    def worker(i, d):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    got[i] = run(d)
                seen[i] = repr(E.cfg())                                           # outside a head's scope: the defaults
            s.synchronize()
        except Exception as e:                                                    # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i, d)) for i, d in enumerate((dens_a, dens_b))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert rel(want[0], want[1]) < 1e-5 and not torch.equal(want[0], want[1])    # two arithmetics, the same density
    assert seen[0] == seen[1] == repr(E.KernelConfig())


def test_rccl_single_rank_runs_the_benchmarks_collectives():
    """VERDICT r3 missing #1: RCCL itself inside the driver-run suite.  A FRESH child process (never a re-exec of this one) builds
    a ONE-rank nccl (= RCCL) process group with the environment bench.py's launcher sets and runs the collectives the benchmark
    issues; a second child runs ``bench.py --force-group``: init_group, barrier + (sum, count) all-reduce per step, MAX of the
    times -- the exact calls of an N > 1 run, on one rank."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dev", "nccl_single_rank.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0 and "nccl ok [3.5, 2.0] 1" in r.stdout, r.stdout + r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "32", "--steps", "1", "--no-legs",
                        "--cpu-batch", "0", "--force-group"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["ranks_seen"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "nccl, 1 rank(s), forced" == d["config"]["process_group"]


def test_packs_of_derived_weights_after_an_optimiser_step_are_fresh():
    """``FlatOptimizer.step`` rewrites the parameters under the tensors' version counters and invalidates the packed
    copies; the batched re-pack must not stamp a pack of a DERIVED tensor (masked MADE weight, L U product: rebuilt from the new
    parameters only on its next use) as fresh.  After a step, the model's elbo equals the elbo computed with every cache dropped."""
    from test_oracle_golden import nsf_model
    from cmf_amd import engine as E
    from cmf_amd.optim import FlatOptimizer
    cfg, schema, shape, dens, sd, sdo, ops = nsf_model("hepmass", 2, (32, 32))
    dens = dens.cuda().train()
    opt = FlatOptimizer(dens.parameters(), opt="adam", lr=1e-2)
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(32, *shape, generator=gen) * 1.5).cuda()
    for _ in range(2):
        opt.zero_grad()
        (-dens.elbo(x, add_offdiagonal_metric_reg=True)["elbo"].mean()).backward()
        opt.step()
    dens.eval()
    with torch.no_grad():
        got = dens.elbo(x, add_offdiagonal_metric_reg=True)["elbo"].clone()
        E.PACKS._store.clear(); E.PACKS._tables.clear(); E.PACKS._refreshed.clear(); E.DERIVED._store.clear()
        want = dens.elbo(x, add_offdiagonal_metric_reg=True)["elbo"]
    assert torch.equal(got, want)


@pytest.mark.parametrize("name", ["c3_mnist_full", "c3_mnist_full_cond"])
def test_full_size_reference_vectors_through_the_grouped_primal_path(name):
    """The full-size fixtures hold B = 2 samples, and batches that are not a multiple of 16 take the plain primal path; repeated
    64-fold (128 samples: 8 sample groups) the same inputs run through the fp16-split primal kernels and must reproduce the
    reference's vectors -- also on the conditioned model (cond(J^T J) ~ 6e2, ScaledTanh gains x 2.5).

    The bound on that model is COMPUTED (conftest.fp64_bound's convention: 3 x the reference arithmetic's own distance from the
    float64 value of the same model on the same input, never below 1e-4): the reference's float32 J^T J of this fixture is itself
    1.3e-4 away from the float64 one (a pre-activation within float32 rounding of zero: the fixture holds the other side of the
    kink), and a primal pass that lands on the float64 side is 1.3e-4 from the fixture while 1.5e-6 from float64.  Both are
    asserted: the HIP result within the bound of the float64 oracle AND of the fixture, and with ``primal="f32"`` (exact fp32
    products, the reference's arithmetic) within 1e-4 of the fixture itself."""
    from cmf_amd import engine as E
    from oracle import cmf_oracle as O
    g, meta, cfg, dens = build(name)
    head = find_head(dens)
    _, _, _, ops, sd = golden_model(meta)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        p64 = O.elbo(sd64, ops, g["x"].double(), noise=g["noise"].double(), add_offdiagonal_metric_reg=True, return_parts=True)["parts"]
    yard = {k: rel(g[k], p64[k]) for k in ("jtj", "logdet")}             # the reference's float32 against float64
    bound = {k: max(1e-4, 3 * v) for k, v in yard.items()}
    assert (bound["jtj"] > 1e-4) == (name == "c3_mnist_full_cond") and bound["jtj"] < 5e-4 and bound["logdet"] == 1e-4, (yard, bound)
    rep = 64
    x = (g["x"] + g["noise"]).repeat(rep, 1, 1, 1).cuda()
    for primal in ("f16x3", "f32"):
        head.kernels = E.KernelConfig(primal=primal)
        with E.timing(lambda n: n.endswith("_primal")) as timer, torch.no_grad():
            out = inner(dens, True).elbo(x, add_offdiagonal_metric_reg=True)["elbo"]
        assert timer.by_name(), "the grouped primal path did not run"
        gr = head.last_gram
        for i in range(0, 2 * rep, 2):                                     # every copy, bit for bit the same
            assert torch.equal(out[i:i + 2], out[0:2])
        jtj, logdet = gr.jtj[0:2].cpu(), gr.logdet[0:2].view(-1, 1).cpu()
        assert rel(jtj, p64["jtj"]) < bound["jtj"] and rel(logdet, p64["logdet"]) < bound["logdet"]
        assert rel(jtj, g["jtj"]) < (bound["jtj"] if primal == "f16x3" else 1e-4) and rel(logdet, g["logdet"]) < 1e-4
        assert rel(out[0:2], g["elbo_0"] if "elbo_0" in g else g["elbo"]) < 1e-4


@pytest.mark.parametrize("H,W,groups", [(28, 28, 4), (14, 14, 4), (32, 32, 2)])
def test_batched_primal_weight_gradients_equal_the_single_launches(H, W, groups):
    """cmf_conv_tangent_wgrad_bf16x3_batched: five problems of one shape (sample groups paired as the two 16-column slices of a
    32-column "sample", the input's own relu: the primal weight gradients of a coupler's hidden convs) in ONE launch against one
    launch each -- and both against float64 autograd."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(H + groups)
    C, HW, nprob = 64, H * W, 5
    B = 16 * groups
    gs = C * HW * 16
    xs = [torch.randn(B, C, H, W, generator=gen) for _ in range(nprob)]
    gys = [torch.randn(B, C, H, W, generator=gen) for _ in range(nprob)]
    xg = [E.primal_regroup(x.cuda(), True) for x in xs]
    gg = [E.primal_regroup(g.cuda(), True) for g in gys]
    single = [torch.zeros(C, C, 3, 3, device="cuda") for _ in range(nprob)]
    batched = [torch.full((C, C, 3, 3), 0.5, device="cuda") for _ in range(nprob)]            # accumulated INTO
    for x, g, dw in zip(xg, gg, single):
        E.conv_tangent_wgrad(x, 0, 2 * gs, HW * 16, 16, g, 0, 2 * gs, HW * 16, 16, dw, 9, groups // 2, C, C, H, W, 32,
                             fmode=E.F_SELF_RELU, x_sl=gs, y_sl=gs)
    E.conv_tangent_wgrad_batched(xg, gg, batched, 2 * gs, HW * 16, 16, 2 * gs, HW * 16, 16, groups // 2, C, C, H, W, 32,
                                 fmode=E.F_SELF_RELU, x_sl=gs, y_sl=gs)
    for x, g, d1, d2 in zip(xs, gys, single, batched):
        w = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
        (F.conv2d(torch.relu(x).double(), w, padding=1) * g.double()).sum().backward()
        assert rel(d1, w.grad) < 5e-5 and rel(d2 - 0.5, w.grad) < 5e-5
        assert rel(d2 - 0.5, d1) < 2e-6                         # the same products, dealt to 48 instead of up to 56 workgroups


def test_batched_channel_sums_equal_the_single_launches():
    """cmf_channel_sum_batched: the bias gradients of several cotangent tensors of one shape (16 samples in the column slots) in ONE
    launch, accumulated INTO their outputs like cmf_channel_sum."""
    from cmf_amd import engine as E
    gen = torch.Generator().manual_seed(2)
    G, C, HW, n = 3, 64, 14 * 14, 5
    ts = [torch.randn(G * C * HW * 16, generator=gen).cuda() for _ in range(n)]
    pn = (C * HW * 16, HW * 16, 16)
    single = [torch.full((C,), 0.25, device="cuda") for _ in range(n)]
    batched = [torch.full((C,), 0.25, device="cuda") for _ in range(n)]
    for t, o in zip(ts, single):
        E.channel_sum(t, *pn, G, C, HW, 16, o)
    E.channel_sum_batched(ts, *pn, G, C, HW, 16, batched)
    for t, a, b in zip(ts, single, batched):
        assert torch.equal(a, b)
        assert rel(a - 0.25, t.view(G, C, HW * 16).double().sum((0, 2))) < 1e-5
