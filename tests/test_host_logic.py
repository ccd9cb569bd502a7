"""Host-side logic that needs no GPU: schema builder, factory / state-dict schema, index maps, loss closures,
weight recipe, refusal of CPU tensors."""
import json

import numpy as np
import pytest
import torch

import cmf_amd
from cmf_amd import schemas
from cmf_amd.recipe import fill_state_dict
from conftest import load_golden, golden_model
from oracle import cmf_oracle as O

CASES = ["c1_sphere", "c1_sphere_d2", "c2a_power", "c2b_hepmass", "mini_mnist", "mini_cifar", "mini_mnist_small", "c3_mnist_full"]


@pytest.mark.parametrize("name", CASES)
def test_state_dict_schema_equals_reference(name):
    """Key names, order, shapes and dtypes of state_dict() equal the reference model's (dumped into the fixture)."""
    g, meta = load_golden(name)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    sd = dens.state_dict()
    assert list(sd) == list(meta["state_dict"])
    for k, v in sd.items():
        assert [list(v.shape), str(v.dtype)] == meta["state_dict"][k], k


def test_mnist_model_size_and_depth():
    g, meta = load_golden("c3_mnist_full")
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config("mnist", latent_dimension=64)), g["x"])
    assert sum(p.numel() for p in dens.parameters()) == 5_983_910
    assert len(dens.state_dict()) == 486
    assert max(k.count(".") for k in dens.state_dict()) >= 30


def test_recipe_is_deterministic_and_loads():
    g, meta = load_golden("mini_mnist")
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    a, b = fill_state_dict(dens.state_dict(), 0), fill_state_dict(dens.state_dict(), 0)
    c = fill_state_dict(dens.state_dict(), 1)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert any(not torch.equal(a[k], c[k]) for k in a)
    dens.load_state_dict(a, strict=True)
    _, _, _, _, sd_ref = golden_model(meta)
    assert all(torch.equal(a[k].float() if a[k].is_floating_point() else a[k], sd_ref[k]) for k in a), "recipe vs fixture rebuild"
    perm = a["module.density.prior.prior.prior.prior." + "prior." * 7 + "density_1.prior.prior.prior.prior.permutation"]
    assert sorted(perm.tolist()) == list(range(392))


def test_schema_shapes_of_baseline_configs():
    s = schemas.get_schema(schemas.get_config("mnist", latent_dimension=64))
    kinds = [l["type"] for l in s]
    assert kinds[:5] == ["dequantization", "scalar-mult", "scalar-add", "logit", "non-square-head"]
    assert kinds.count("acl") == 10 + 10 and kinds.count("squeeze") == 1 and kinds.count("split") == 1
    assert s[1]["value"] == pytest.approx((1 - 2e-6) / 256)
    s5 = schemas.get_schema(schemas.get_config("cifar10", latent_dimension=128, hutchinson_samples=4))
    assert s5[4]["log_jacobian_method"] == "hutch_with_cg" and s5[4]["hutchinson_samples"] == 4 and s5[2]["value"] == 0.05
    s1 = schemas.get_schema(schemas.get_config("sphere", latent_dimension=3))
    assert [l["type"] for l in s1] == ["non-square-head", "flatten"] + ["acl"] * 5 + ["non-square-base", "affine"]
    nsf = [l["type"] for l in schemas.get_schema({**schemas.get_config("power"), "prior": "nsf"})]
    tail = nsf[nsf.index("non-square-base") + 1:]                       # schemas.py:87-103, :586-626 without 'normalise'
    assert tail == ["flatten"] + ["rand-channel-perm", "linear", "nsf-ar"] * 5 + ["rand-channel-perm", "linear"]
    with pytest.raises(ValueError):
        schemas.get_schema({**schemas.get_config("power"), "prior": "standard-normal"})
    with pytest.raises(KeyError):
        schemas.get_config("power", no_such_key=1)


def test_invalid_method_raises_like_reference():
    from cmf_amd.densities import NonSquareHeadDensity
    with pytest.raises(ValueError, match="not a valid Jacobian calculation method"):
        NonSquareHeadDensity(None, 1, "lu", (3,), "normal")


def _find(dens, cls):
    m = dens
    while type(m).__name__ != cls:
        mods = m._modules
        m = mods.get("module") or mods.get("density") or mods.get("prior") or mods.get("density_1")
    return m


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar", "c2b_hepmass", "c1_sphere"])
def test_index_maps_reproduce_reference_layer_semantics(name):
    """Apply each layer's index maps with numpy on CPU and compare with the oracle's tensor formulation."""
    from cmf_amd.bijections import AffineCouplingBijection, Squeeze2dBijection
    g, meta = load_golden(name)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    head = _find(dens, "NonSquareHeadDensity")
    prog = head.program
    _, _, _, ops, sd = golden_model(meta)
    pre, _, flow_ops, base, prior_ops = O.split_ops(ops)
    gen = torch.Generator().manual_seed(0)
    acl_ops = [o for o in flow_ops if o["kind"] == "acl"]
    acl_mods = [m for m in prog.layers if isinstance(m, AffineCouplingBijection)]
    assert len(acl_ops) == len(acl_mods)
    for op, m in zip(acl_ops, acl_mods):
        shape = op["x_shape"]
        z = torch.randn(2, *shape, generator=gen)
        y = torch.randn(2, op["cout"], *shape[1:], generator=gen)
        zi, si, ti = (m._maps._host[k] for k in ("zi", "si", "ti"))
        # decode with maps: x = z * exp(-s) - t on modified elements
        out = z.reshape(2, -1).clone()
        yf = y.reshape(2, -1)
        out[:, zi] = out[:, zi] * torch.exp(-yf[:, si]) - yf[:, ti]
        t, s = O._chunk(y)
        if op["mask_type"] == "checkerboard":
            mk = O.checkerboard_mask(shape, op["reverse"], z)
            want = mk * z + (1 - mk) * (z * torch.exp(-s) - t)
            assert torch.equal(m.mask, mk)
        else:
            zp, zm = O._cw_split(op, z)
            want = O._cw_combine(op, zp, zm * torch.exp(-s) - t)
            off, step, n = m._pass
            C = shape[0]
            assert torch.equal(z[:, off:off + step * n:step] if step > 1 or n > 1 else z[:, off:off + 1], zp)
        assert torch.allclose(out.reshape(want.shape), want, atol=1e-6)
        assert m.net.kind == ("resnet" if op["net"] == "resnet" else "mlp")
    for m in prog.layers:
        if isinstance(m, Squeeze2dBijection):
            x = torch.randn(2, *m.x_shape, generator=gen)
            zz = O.squeeze_x_to_z(x, m.factor)
            assert torch.equal(x.reshape(2, -1)[:, m._maps._host["x2z"]].reshape(zz.shape), zz)
            assert torch.equal(zz.reshape(2, -1)[:, m._maps._host["z2x"]].reshape(x.shape), x)
    tail = prog.tail
    z_low = torch.randn(2, tail.latent_dimension, generator=gen)
    idx = tail.scatter_index("cpu").long()
    dense = torch.where(idx >= 0, z_low[:, idx.clamp_min(0)], torch.zeros(()))
    sdk = {base["prefix"] + "permutation": tail.permutation, base["prefix"] + "inverse_permutation": tail.inverse_permutation}
    assert torch.equal(dense.reshape(2, *tail.x_shape), O.tail_scatter(sdk, base, z_low))
    h = torch.randn(2, *tail.x_shape, generator=gen)
    assert torch.equal(h.flatten(1)[:, tail.gather_index("cpu").long()], O.tail_gather(sdk, base, h))


class _Recorder:
    def __init__(self):
        self.calls = []

    def elbo(self, x, **kw):
        self.calls.append(kw)
        return {"elbo": torch.arange(4.).view(4, 1)}


def test_train_metric_closures_pass_reference_kwargs():
    """non_square_helpers.py:31-135: kwarg names / values handed to density.elbo and the warm-up ramp."""
    cfg = {"m_flow": False, "likelihood_warmup": True, "likelihood_warmup_start": 25, "likelihood_warmup_end": 50,
           "g_kk_loss": False, "g_ij_loss": True, "latent_dimension": 64, "elbo_regularization_param": 2.0,
           "metric_regularization_param": 3.0}
    fn, intro, early = cmf_amd.get_non_square_train_metrics(cfg)
    assert (intro, early) == (25, 50)
    rec = _Recorder()
    for epoch, w in [(0, 0.0), (25, 0.0), (30, 0.2), (50, 1.0), (80, 1.0)]:
        out = fn(rec, None, epoch)
        kw = rec.calls[-1]
        assert kw == {"likelihood_wt": pytest.approx(w * 2.0), "metric_wt": pytest.approx(w * 3.0), "add_reconstruction": True,
                      "add_diagonal_metric_reg": False, "add_offdiagonal_metric_reg": True}
        assert float(out["loss"]) == pytest.approx(-1.5)
    fn, _, _ = cmf_amd.get_non_square_train_metrics({**cfg, "g_ij_loss": False, "g_kk_loss": True})
    fn(rec, None, 60)
    assert rec.calls[-1]["add_diagonal_metric_reg"] is True and rec.calls[-1]["add_offdiagonal_metric_reg"] is False
    fn, intro, early = cmf_amd.get_non_square_train_metrics({**cfg, "g_ij_loss": False, "likelihood_warmup": False})
    fn(rec, None, 3)
    assert rec.calls[-1] == {"likelihood_wt": 1.0, "add_reconstruction": True} and (intro, early) == (0, 0)
    with pytest.raises(AssertionError):
        cmf_amd.get_non_square_train_metrics({**cfg, "g_kk_loss": True})
    with pytest.raises(AssertionError):
        cmf_amd.get_non_square_train_metrics({**cfg, "latent_dimension": 1})
    # m-flow alternates objectives every epoch: odd epochs carry the likelihood, even ones the reconstruction
    fn, intro, early = cmf_amd.get_non_square_train_metrics({**cfg, "m_flow": True, "g_ij_loss": False})
    assert (intro, early) == (50, 100)
    fn(rec, None, 101); assert rec.calls[-1] == {"likelihood_wt": 1.0, "add_reconstruction": False}
    fn(rec, None, 100); assert rec.calls[-1] == {"likelihood_wt": 0, "add_reconstruction": True}


def test_parameter_groups():
    g, meta = load_golden("c1_sphere")
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config("sphere", latent_dimension=3)), g["x"])
    groups = cmf_amd.get_non_square_parameters(dens, m_flow=False)
    assert sum(p.numel() for p in groups[0]) == 840
    mf = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config("sphere", latent_dimension=2, m_flow=True)), g["x"])
    rec, lik = (list(gr) for gr in cmf_amd.get_non_square_parameters(mf, m_flow=True))
    assert len(lik) == 2 and sum(p.numel() for p in lik) == 4              # the affine prior's shift + log_scale
    assert sum(p.numel() for p in rec) + 4 == sum(p.numel() for p in mf.parameters())


def test_product_path_has_no_cpu_fallback_and_no_oracle_import():
    import os
    import re
    g, meta = load_golden("c1_sphere")
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config("sphere", latent_dimension=3)), g["x"]).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        dens.elbo(g["x"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in os.listdir(os.path.join(root, "cmf_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(root, "cmf_amd", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    # repo-wide: only tests/, smoke() (__graft_entry__.py) and bench.py's cpu_baseline leg touch oracle/
    allowed = {os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")}
    for d, _, files in os.walk(root):
        rel = os.path.relpath(d, root)
        if rel.split(os.sep)[0] in ("tests", "oracle", ".git", "gpurun_out", ".pytest_cache"):
            continue
        for f in files:
            path = os.path.join(d, f)
            if f.endswith(".py") and path not in allowed:
                assert not re.search(r"^\s*(from|import)\s+oracle", open(path).read(), flags=re.M), path


def test_dequantization_mutates_caller_tensor_like_reference():
    """wrapper.py:30 adds the noise in place; we keep that (checked on CPU up to the point the GPU is needed)."""
    from cmf_amd.densities import DequantizationDensity, Density

    class Sink(Density):
        def _elbo(self, x, **kw):
            return {"elbo": x.sum().view(1, 1)}

    x = torch.zeros(2, 1, 4, 4)
    DequantizationDensity(Sink()).elbo(x)
    assert float(x.min()) >= 0 and float(x.max()) < 1 and float(x.abs().sum()) > 0


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """f4: files in the reference trainer's format (trainer.py:362-400, writer.py:105-126) written and read back; the
    module_state_dict keys are the reference's own (golden meta), so either side can read the other's file."""
    import math
    from cmf_amd import checkpoint as ck
    g, meta = load_golden("mini_mnist")
    build = lambda: cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    a, b = build(), build()
    from cmf_amd.recipe import fill_state_dict
    a.load_state_dict(fill_state_dict(a.state_dict(), seed=3))
    opt = torch.optim.Adam([p for p in a.parameters() if p.requires_grad], lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 5)
    path = ck.write_checkpoint(str(tmp_path), "latest", a, [opt], [sched], epoch=7, iteration=123, best_valid_loss=1.5,
                               num_bad_valid_epochs=2)
    assert path == str(tmp_path / "checkpoints" / "latest.pt") and not (tmp_path / "checkpoints" / "latest.pt.tmp").exists()
    raw = torch.load(path, weights_only=False)
    assert tuple(raw) == ck.KEYS and list(raw["module_state_dict"]) == list(meta["state_dict"])
    opt_b = torch.optim.Adam([p for p in b.parameters() if p.requires_grad], lr=1e-3)
    out = ck.load_checkpoint(str(tmp_path), "latest", b, [opt_b], [torch.optim.lr_scheduler.StepLR(opt_b, 5)])
    assert (out["epoch"], out["iteration"], out["best_valid_loss"], out["num_bad_valid_epochs"]) == (7, 123, 1.5, 2)
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k
    # a file whose keys lack nn.DataParallel's ``module.`` prefix (a model built the other way), and the old single-dict spelling
    assert all(k.startswith("module.") for k in raw["module_state_dict"])
    raw["module_state_dict"] = {k[len("module."):]: v for k, v in raw["module_state_dict"].items()}
    raw["opt_state_dict"] = raw.pop("opt_state_dicts")[0]
    raw["lr_scheduler_state_dict"] = raw.pop("lr_scheduler_state_dicts")[0]
    torch.save(raw, path)
    c = build()
    opt_c = torch.optim.Adam([p for p in c.parameters() if p.requires_grad], lr=1e-3)
    ck.load_checkpoint(str(tmp_path), "latest", c, [opt_c], [torch.optim.lr_scheduler.StepLR(opt_c, 5)])
    assert all(torch.equal(v, c.state_dict()[k]) for k, v in a.state_dict().items())
    assert math.isinf(ck.write_checkpoint.__defaults__[-2])


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar", "c2b_hepmass", "c1_sphere"])
def test_structural_zero_plan_matches_the_oracles_data(name):
    """``FlowProgram.zero_in`` (a static walk over the layer list) marks exactly the coupling layers whose network input the
    ORACLE finds identically zero -- primal and every tangent column -- in a decode sweep from a random latent: the coupler
    right behind ``SplitDensity.pad_inputs`` (split.py:50-52) whose pass-through half is the padded one (acl.py:148-160)."""
    from cmf_amd.bijections import AffineCouplingBijection, SplitChannelwiseAffineCouplingBijection
    from cmf_amd.densities import SplitDensity
    g, meta = load_golden(name)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    prog = _find(dens, "NonSquareHeadDensity").program
    _, _, _, ops, sd = golden_model(meta)
    _, _, flow_ops, base, _ = O.split_ops(ops)
    gen = torch.Generator().manual_seed(3)
    d = base["d"]
    z = torch.randn(2, d, generator=gen)
    h, hv = O.tail_scatter(sd, base, z), O.tail_scatter(sd, base, torch.eye(d).expand(2, d, d))
    zero_ops = []                                             # position among the acl ops, decode order
    k = 0
    with torch.no_grad():
        for op in reversed(flow_ops):
            if op["kind"] == "acl":
                if op["mask_type"] == "checkerboard":
                    m = sd[op["prefix"] + "mask"]
                    net_in, net_v = m * h, m * hv
                else:
                    net_in, net_v = O._cw_split(op, h)[0], O._cw_split(op, hv, 2)[0]
                if float(net_in.abs().max()) == 0.0 and float(net_v.abs().max()) == 0.0:
                    zero_ops.append(k)
                k += 1
                h, hv = O.acl_jvp_multi(sd, op, h, hv)
            elif op["kind"] == "flatten":
                h, hv = h.reshape(h.shape[0], *op["x_shape"]), hv.reshape(*hv.shape[:2], *op["x_shape"])
            elif op["kind"] == "squeeze":
                h, hv = O.squeeze_z_to_x(h, op["factor"]), O.squeeze_z_to_x(hv, op["factor"])
            elif op["kind"] == "split":
                h, hv = torch.cat((h, torch.zeros_like(h)), 1), torch.cat((hv, torch.zeros_like(hv)), 2)
    order = [i for i, _ in prog._decode_order()]
    acl_pos = {i: n for n, i in enumerate(i for i in order if isinstance(prog.layers[i], AffineCouplingBijection))}
    assert sorted(acl_pos[i] for i in prog.zero_in) == zero_ops
    if name.startswith("mini"):                               # image schemas (schemas.py:399-412): exactly one, right behind the pad
        (i,) = prog.zero_in
        assert isinstance(prog.layers[i], SplitChannelwiseAffineCouplingBijection) and prog.layers[i].reverse_mask
        assert isinstance(prog.layers[i + 1], SplitDensity)   # decode order runs the list backwards
    else:
        assert not prog.zero_in


@pytest.mark.parametrize("name", ["mini_mnist", "mini_cifar", "c3_mnist_full"])
def test_seed_column_plan_matches_the_oracles_data(name):
    """``FlowProgram._seed_columns``: the Jacobian columns whose tangent the FIRST decoded coupler's network sees as non-zero are
    exactly the ones the plan packs -- checked on the oracle's own tangents (tail scatter of the identity, times the layer's mask)."""
    from cmf_amd.bijections import AffineCouplingBijection
    g, meta = load_golden(name)
    dens = cmf_amd.get_density(cmf_amd.get_schema(cmf_amd.get_config(meta["dataset"], **meta["overrides"])), g["x"])
    prog = _find(dens, "NonSquareHeadDensity").program
    _, _, _, ops, sd = golden_model(meta)
    dens.load_state_dict({k: v for k, v in sd.items()}, strict=False)          # the fixture's permutation buffers
    _, _, flow_ops, base, _ = O.split_ops(ops)
    d = base["d"]
    hv = O.tail_scatter(sd, base, torch.eye(d).expand(1, d, d))                 # (1, d, *x_shape): column k's seed tangent
    op = [o for o in flow_ops if o["kind"] == "acl"][-1]                        # first in decode order
    assert op["mask_type"] == "checkerboard"
    seen = (sd[op["prefix"] + "mask"] * hv).flatten(2).abs().amax(2)[0] > 0     # columns with a non-zero network input
    plan = prog._seed_columns("cpu")
    first = prog.layers[-1]
    assert isinstance(first, AffineCouplingBijection)
    cols = [k for k in range(d) if bool(seen[k])]
    if plan is None:
        assert (len(cols) + 15) // 16 * 16 >= (d + 15) // 16 * 16 or first.net.kind != "resnet"
        return
    colmap = plan["colmap"].numpy()
    assert [k for k in range(d) if colmap[k] >= 0] == cols and plan["n"] == len(cols) and plan["nc"] == (len(cols) + 15) // 16 * 16
    assert sorted(colmap[colmap >= 0].tolist()) == list(range(len(cols)))
    col_of = plan["col_of"].numpy()
    pos = prog.tail.permutation[:d].numpy()
    assert all(col_of[pos[k]] == colmap[k] for k in cols) and int((col_of >= 0).sum()) == len(cols)
