import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Load a fixture written by oracle/make_golden.py -> (arrays dict of torch tensors, meta dict)."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return arrs, meta


def golden_model(meta, dtype=torch.float32):
    """Rebuild (schema, x_shape, oracle ops, recipe state dict) for a fixture without the reference:
    the state-dict key/shape list is stored in the fixture, values come from the recipe."""
    from cmf_amd import schemas
    from cmf_amd.recipe import apply_gain, recipe_tensor
    from oracle import cmf_oracle as O

    cfg = schemas.get_config(meta["dataset"], **meta["overrides"])
    schema = schemas.get_schema(cfg)
    x_shape = schemas.DATA_SHAPES[meta["dataset"]]
    ops = O.compile_schema(schema, x_shape)
    shapes = {k: tuple(v[0]) for k, v in meta["state_dict"].items()}
    sd = {}
    for k, (shape, dt) in meta["state_dict"].items():
        tdt = getattr(torch, dt.replace("torch.", ""))
        t = apply_gain(k, recipe_tensor(k, shape, tdt, meta["recipe_seed"], shapes), meta.get("recipe_gain"))
        if t is None:
            t = structural_buffer(k, tuple(shape), ops, tdt)
        if t.is_floating_point():
            t = t.to(dtype)
        sd[k] = t.reshape(shape)
    return cfg, schema, x_shape, ops, sd


def structural_buffer(key, shape, ops, dtype):
    """Constructor-valued buffers the recipe leaves alone: checkerboard masks (acl.py:68-78),
    tail mask (non_square.py:377), Gaussian mean/stddev (factory.py:196-201)."""
    from oracle import cmf_oracle as O
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "mean":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "stddev":
        return torch.ones(shape, dtype=dtype)
    if leaf == "mask" and len(shape) == 1:
        for op in ops:
            if op["kind"] == "base" and op["prefix"] + "mask" == key:
                return torch.arange(shape[0]) < op["d"]
    if leaf == "mask":
        for op in ops:
            if op["kind"] == "acl" and op["prefix"] + "mask" == key:
                return O.checkerboard_mask(shape, op["reverse"], torch.zeros(1, dtype=dtype))
    raise KeyError(key)


#: fixture families (oracle/make_golden.py CASES)
SMALL = ["c1_sphere", "c1_sphere_d2", "c2a_power", "c2b_hepmass", "mini_mnist", "mini_cifar", "mini_mnist_small"]
#: the small models with the recipe's gain raised until the REFERENCE reports cond(J^T J) ~ 1e2 ... 4e3
COND = ["mini_mnist_cond1e2", "mini_mnist_cond1e3", "mini_cifar_cond1e2", "mini_cifar_cond1e3", "c2b_hepmass_cond1e2",
        "c2b_hepmass_cond1e3", "c2b_hepmass_cond4e3"]
#: full-size models: (B, 1) outputs and J^T J only
FULL = ["c3_mnist_full", "c3_mnist_full_cond", "c5_cifar_full"]


def kink_tolerance(g, base=1e-4):
    """End-to-end tolerance of a fixture, COMPUTED from the reference's own data: ``base`` unless the float32 reference itself
    moves by more when its latent moves by ~1e-6 relative per component (``logdet_pert`` / ``l1_off_pert``, written by
    oracle/make_golden.py: seeded random draws) -- i.e. unless the fixture point sits within rounding of a relu kink, where any fp32 encode chain may land on the
    other side.  Then: 3 x that movement (max-norm over the batch, like the assertions it feeds)."""
    if "logdet_pert" not in g:
        return base
    ld = g["logdet"].double().reshape(-1)
    off = (g["jtj"].abs().sum((1, 2)) - torch.diagonal(g["jtj"], dim1=1, dim2=2).abs().sum(1)).double()
    # in the max-norm the end-to-end assertions of test_parts_match_reference_vectors use (test_gpu_parity.rel)
    move = max(float((g["logdet_pert"].double() - ld).abs().max() / ld.abs().max()),
               float((g["l1_off_pert"].double() - off).abs().max() / off.abs().max()))
    return max(base, 3.0 * move)


def fp64_bound(fp64, ref32, pert=None, extra=None, rel=1e-4, k=3.0, abs_floor=None):
    """Per-sample admissible |value - fp64|: max(rel * |fp64| (floored at 1 % of the batch's largest magnitude), k x yardstick,
    abs_floor) with yardstick_b = max(|reference_fp32 - fp64|, reference_fp32's own movement under the +-1e-6 latent
    perturbation) -- the reference arithmetic's own distance from the exact value at that sample.
    ``abs_floor`` (per sample) is the rounding floor of a quantity formed by cancellation: the g_ij sum of a sample whose
    off-diagonal mass is 0.6 % of its diagonal (mini_mnist_cond1e2, sample 0: 2.4 against 393) cannot be known to 1e-4 of ITSELF
    by any fp32 Gram matrix -- float32 dot products over D rows carry ~sqrt(D) 2^-24 ~ 2e-6 of |J_i| |J_j| -- so its floor is
    2e-6 x (sum_ij |G_ij|), the same backward-error scale for every implementation."""
    fp64, ref32 = fp64.double().reshape(-1), ref32.double().reshape(-1)
    yard = (ref32 - fp64).abs()
    if pert is not None:
        yard = torch.maximum(yard, (pert.double().reshape(-1, ref32.numel()) - ref32).abs().max(0).values)
    if extra is not None:
        yard = yard + extra.double().reshape(-1)
    floor = 0.01 * fp64.abs().max()
    bound = torch.maximum(rel * fp64.abs().clamp_min(floor), k * yard)
    if abs_floor is not None:
        bound = torch.maximum(bound, abs_floor.double().reshape(-1))
    return bound, yard


@pytest.fixture(scope="session")
def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f != "jitter_retry.npz")


@pytest.fixture
def kernel_scope():
    """``kernel_scope(tangent=..., primal=...)``: run the rest of the test under that engine.KernelConfig (thread-local scope)."""
    import contextlib
    from cmf_amd import engine as E
    with contextlib.ExitStack() as stack:
        yield lambda **kw: stack.enter_context(E.scope(**kw))
