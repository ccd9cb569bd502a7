"""Deterministic, build-owned weight recipe: ``w = f(seed, key, shape)``.

Fills every entry of a ``state_dict`` (ours or the reference's -- the key schema is
identical) from a per-key numpy PCG64 stream, so golden fixtures
need not store multi-MB weights: the fixture generator loads the recipe into the
reference with ``load_state_dict(strict=True)`` and the tests load the same recipe into
the HIP path and the oracle.

Distribution choices follow PyTorch's defaults for ``nn.Linear``/``nn.Conv2d``
(uniform +-1/sqrt(fan_in), as used by the reference's constructors
``cmf/models/components/networks.py:116-224``) with non-trivial values for the
parameters the reference initialises to constants (ScaledTanh ``weights``/``bias``,
``networks.py:96-101``; AffineBijection ``shift``/``log_scale``, ``affine.py:21-22``) so
that those code paths are exercised by parity tests.
"""
import zlib

import numpy as np
import torch

__all__ = ["fill_state_dict", "recipe_tensor", "apply_gain"]


def _rng(key, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 1000003 * (seed + 1)) % (2 ** 63)))


def recipe_tensor(key, shape, dtype, seed, sibling_shapes=None):
    """Return the recipe value for one state-dict entry, or ``None`` to keep the
    constructor's value (structural buffers: masks, Gaussian mean/stddev)."""
    shape = tuple(shape)
    rng = _rng(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "permutation":
        return torch.from_numpy(rng.permutation(shape[0]).astype(np.int64))
    if leaf == "inverse_permutation":
        perm = _rng(key[: -len("inverse_permutation")] + "permutation", seed).permutation(shape[0])
        return torch.from_numpy(np.argsort(perm).astype(np.int64))
    if leaf in ("mask", "mean", "stddev", "degrees"):
        return None
    if leaf in ("lower_entries", "upper_entries"):      # LULinear strict triangles (nsf prior): away from the identity init
        return torch.from_numpy(rng.uniform(-0.2, 0.2, shape)).to(dtype)
    if leaf == "unconstrained_upper_diag":              # softplus(.) + eps in (0.55, 1.5)
        return torch.from_numpy(rng.uniform(-0.3, 1.2, shape)).to(dtype)
    if leaf == "_fixed_samples":
        return torch.from_numpy(rng.standard_normal(shape)).to(dtype)
    if leaf in ("shift", "log_scale"):          # AffineBijection (2-D prior)
        return torch.from_numpy(rng.uniform(-0.3, 0.3, shape)).to(dtype)
    if leaf == "weights":                       # ScaledTanh2dModule scale
        return torch.from_numpy(rng.uniform(0.4, 0.8, shape)).to(dtype)
    if leaf == "bias" and len(shape) == 3:      # ScaledTanh2dModule offset (C,1,1)
        return torch.from_numpy(rng.uniform(-0.05, 0.05, shape)).to(dtype)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return torch.from_numpy(rng.uniform(-b, b, shape)).to(dtype)
    if leaf == "bias":
        wshape = (sibling_shapes or {}).get(key[:-4] + "weight")
        fan_in = int(np.prod(wshape[1:])) if wshape is not None else shape[0]
        b = 1.0 / np.sqrt(fan_in)
        return torch.from_numpy(rng.uniform(-b, b, shape)).to(dtype)
    raise KeyError(f"recipe has no rule for state-dict key {key!r} of shape {shape}")


def apply_gain(key, t, gain):
    """``gain`` = {leaf name: multiplier} for floating-point recipe tensors -- the knob the ill-conditioned parity fixtures
    turn: ``{"weights": 3.0}`` triples every ScaledTanh scale (the bound of an image coupler's log-scale),
    ``{"weight": 1.5}`` every conv / linear weight.  Only the low-dimensional prior flows (keys holding no image / tabular
    coupler above the base) are left alone when the leaf is given as ``"flow.weight"``-style entries; plain leaves match all."""
    if not gain or t is None or not t.is_floating_point():
        return t
    leaf = key.rsplit(".", 1)[-1]
    g = gain.get(leaf)
    return t if g is None else t * float(g)


def fill_state_dict(state_dict, seed=0, gain=None):
    """Return a new state dict with the same keys/shapes/dtypes filled by the recipe (``gain``: see ``apply_gain``)."""
    shapes = {k: tuple(v.shape) for k, v in state_dict.items()}
    out = {}
    for k, v in state_dict.items():
        t = apply_gain(k, recipe_tensor(k, v.shape, v.dtype, seed, shapes), gain)
        out[k] = v.detach().clone() if t is None else t.reshape(v.shape).to(v.dtype)
    return out
