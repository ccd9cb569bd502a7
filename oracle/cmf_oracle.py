"""CPU oracle for the non-square-flow log-density path of k-flouris/cmf.

TEST INFRASTRUCTURE ONLY.  This file is a functional (state-dict + schema) CPU restatement,
in plain PyTorch, of the algorithm in the reference's
``cmf/models/components/densities/non_square.py`` and the modules it calls.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``cmf_amd`` never does.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference itself (in the
build container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.  The Hutchinson solve
(``gpytorch.utils.linear_cg`` @ fc2053b, un-vendored) is the one exception: see
``hutchinson_surrogate`` -- parity unpinned for the CG iterates, pinned for J^T J eps.  The NSF prior layers
(``nsf_x_to_z``; jrmcornish/nsf @ 8e3fe75, un-vendored) are the other: parity unpinned, restated from the published algorithm.

Two flavours of the Jacobian assembly are provided:
  * ``jtj_ref_equivalent``: one full decode per Jacobian column with the primal of every
    coupler network recomputed per column -- exactly what the reference executes
    (non_square.py:298-311 -> :322-329 -> jvp_layers.py:49-64).  This is the timing
    baseline (``bench.py`` cpu_baseline, kind "port").
  * ``jtj_batched``: all d tangent columns pushed at once (the correctness model of the
    HIP kernels).
All functions are dtype-generic: feed float64 state/inputs for an fp64 evaluation.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# schema -> flat op list (mirrors cmf/models/factory.py:55-162 get_density[_recursive])
# --------------------------------------------------------------------------------------


def compile_schema(schema, x_shape):
    """Flatten the nested Density chain the reference factory would build into a list of
    ops, each carrying the state-dict key prefix of the module that owns its tensors."""
    x_shape = tuple(int(s) for s in x_shape)
    prefix = "module." if x_shape[0] != 2 else ""      # factory.py:76-81 DataParallel wrap
    shape = x_shape
    ops = []
    for layer in schema:
        t = layer["type"]
        if t == "dequantization":                      # factory.py:98-104, wrapper.py:28-30
            ops.append({"kind": "dequant"})
            prefix += "density."
        elif t == "non-square-head":                   # factory.py:120-145
            ops.append({"kind": "head", "regularization_param": layer["regularization_param"],
                        "log_jacobian_method": layer["log_jacobian_method"], "x_shape": shape,
                        "latent_dimension": layer["latent_dimension"]})
            prefix += "prior."
        elif t == "split":                             # factory.py:106-116
            ops.append({"kind": "split", "in_shape": shape})
            shape = (shape[0] // 2, *shape[1:])
            prefix += "density_1."
        elif t == "non-square-base":                   # factory.py:147-156
            ops.append({"kind": "base", "prefix": prefix, "x_shape": shape, "d": layer["latent_dimension"]})
            shape = (layer["latent_dimension"],)
            prefix += "prior."
        else:                                          # factory.py:166-174 BijectionDensity
            op = {"kind": t, "prefix": prefix + "bijection.", "x_shape": shape}
            if t == "flatten":
                shape = (int(np.prod(shape)),)
            elif t == "squeeze":
                f = layer["factor"]
                op["factor"] = f
                shape = (shape[0] * f * f, shape[1] // f, shape[2] // f)
            elif t in ("scalar-mult", "scalar-add"):
                op["value"] = layer["value"]
            elif t == "logit":
                pass
            elif t == "affine":
                assert not layer["per_channel"]
            elif t == "acl":
                op.update(_acl_spec(layer, shape))
            elif t in ("rand-channel-perm", "linear"):    # factory.py:287-292 (nsf prior only: schemas.py:87-103)
                assert len(shape) == 1
            elif t == "nsf-ar":                           # factory.py:304-314
                assert len(shape) == 1 and layer["activation"] == "relu" and layer["dropout_probability"] == 0.
                op.update(hidden=layer["num_hidden_channels"], blocks=layer["num_hidden_layers"], bins=layer["num_bins"],
                          tail_bound=layer["tail_bound"])
            else:
                raise ValueError(f"oracle: layer type {t!r} is outside the hot path")
            op["z_shape"] = shape
            ops.append(op)
            prefix += "prior."
    ops.append({"kind": "gaussian", "prefix": prefix, "shape": shape})   # factory.py:92-94
    return ops


def _acl_spec(layer, shape):
    """factory.py:358-393 + acl.py:81-99,169-214: passthrough count and coupler net sizes."""
    C = shape[0]
    mt, rev = layer["mask_type"], layer["reverse_mask"]
    net = layer["coupler"]["shift_log_scale_net"]
    assert not layer["coupler"]["independent_nets"] and layer["num_u_channels"] == 0
    if mt == "checkerboard":
        cin, cmod = C, C
    else:
        npass = C // 2 if mt == "split-channel" else (C + 1) // 2
        if rev:
            npass = C - npass
        cin, cmod = npass, C - npass
    spec = {"mask_type": mt, "reverse": rev, "net": net["type"], "hidden": list(net["hidden_channels"]),
            "cin": cin, "cout": 2 * cmod}
    if net["type"] == "mlp":
        assert net["activation"] == "tanh"
    else:
        assert net["type"] == "resnet" and not net.get("batchnorm", True)
    return spec


# --------------------------------------------------------------------------------------
# coupler networks: primal forward and forward-mode tangents
# --------------------------------------------------------------------------------------


def _net_keys(op):
    return op["prefix"] + "coupler.shift_log_scale_net."


def net_forward(sd, op, x):
    """networks.py:206-224 (get_mlp) / :116-161 (get_resnet) + :96-106 ScaledTanh2dModule."""
    p = _net_keys(op)
    if op["net"] == "mlp":
        h = x
        n = len(op["hidden"])
        for i in range(n):
            h = torch.tanh(F.linear(h, sd[f"{p}{2*i}.weight"], sd[f"{p}{2*i}.bias"]))
        return F.linear(h, sd[f"{p}{2*n}.weight"], sd[f"{p}{2*n}.bias"])
    n = len(op["hidden"])
    h = F.conv2d(x, sd[p + "module.0.weight"], None, padding=1)
    for i in range(1, n + 1):                      # ResidualBlock.forward networks.py:50-60
        o = F.conv2d(torch.relu(h), sd[f"{p}module.{i}.conv1.weight"], sd[f"{p}module.{i}.conv1.bias"], padding=1)
        o = F.conv2d(torch.relu(o), sd[f"{p}module.{i}.conv2.weight"], sd[f"{p}module.{i}.conv2.bias"], padding=1)
        h = o + h
    o = F.conv2d(torch.relu(h), sd[f"{p}module.{n+2}.weight"], sd[f"{p}module.{n+2}.bias"])
    return sd[p + "weights"] * torch.tanh(o) + sd[p + "bias"]


def net_jvp(sd, op, x, v):
    """Forward-mode tangent with the primal recomputed alongside, one tangent per sample:
    jvp_layers.py:38-64 rules, networks.py:24-32, :62-79, :108-113.  x, v: (N, ...)."""
    p = _net_keys(op)
    if op["net"] == "mlp":
        h, hv = x, v
        n = len(op["hidden"])
        for i in range(n):
            W = sd[f"{p}{2*i}.weight"]
            h = torch.tanh(F.linear(h, W, sd[f"{p}{2*i}.bias"]))
            hv = (1 - h ** 2) * F.linear(hv, W)
        W = sd[f"{p}{2*n}.weight"]
        return F.linear(h, W, sd[f"{p}{2*n}.bias"]), F.linear(hv, W)
    n = len(op["hidden"])
    W = sd[p + "module.0.weight"]
    h, hv = F.conv2d(x, W, None, padding=1), F.conv2d(v, W, None, padding=1)
    for i in range(1, n + 1):
        W1, b1 = sd[f"{p}module.{i}.conv1.weight"], sd[f"{p}module.{i}.conv1.bias"]
        W2, b2 = sd[f"{p}module.{i}.conv2.weight"], sd[f"{p}module.{i}.conv2.bias"]
        o, ov = F.conv2d(torch.relu(h), W1, b1, padding=1), F.conv2d((h > 0) * hv, W1, None, padding=1)
        o, ov = F.conv2d(torch.relu(o), W2, b2, padding=1), F.conv2d((o > 0) * ov, W2, None, padding=1)
        h, hv = o + h, ov + hv
    W, b = sd[f"{p}module.{n+2}.weight"], sd[f"{p}module.{n+2}.bias"]
    o, ov = F.conv2d(torch.relu(h), W, b), F.conv2d((h > 0) * hv, W, None)
    th = torch.tanh(o)
    return sd[p + "weights"] * th + sd[p + "bias"], sd[p + "weights"] * ((1 - th ** 2) * ov)


def net_jvp_multi(sd, op, x, V):
    """All tangent columns at once.  x: (B, ...), V: (B, d, ...).  The primal is computed
    once; activation derivatives are shared across the d columns of a sample."""
    B, d = V.shape[:2]
    p = _net_keys(op)

    def lin(t, W):
        return F.linear(t, W)

    def cnv(t, W, pad):
        s = t.shape
        return F.conv2d(t.reshape(B * d, *s[2:]), W, None, padding=pad).reshape(B, d, W.shape[0], *s[3:])

    if op["net"] == "mlp":
        h, hv = x, V
        n = len(op["hidden"])
        for i in range(n):
            W = sd[f"{p}{2*i}.weight"]
            h = torch.tanh(F.linear(h, W, sd[f"{p}{2*i}.bias"]))
            hv = (1 - h ** 2).unsqueeze(1) * lin(hv, W)
        W = sd[f"{p}{2*n}.weight"]
        return F.linear(h, W, sd[f"{p}{2*n}.bias"]), lin(hv, W)
    n = len(op["hidden"])
    W = sd[p + "module.0.weight"]
    h, hv = F.conv2d(x, W, None, padding=1), cnv(V, W, 1)
    for i in range(1, n + 1):
        W1, b1 = sd[f"{p}module.{i}.conv1.weight"], sd[f"{p}module.{i}.conv1.bias"]
        W2, b2 = sd[f"{p}module.{i}.conv2.weight"], sd[f"{p}module.{i}.conv2.bias"]
        o, ov = F.conv2d(torch.relu(h), W1, b1, padding=1), cnv((h > 0).unsqueeze(1) * hv, W1, 1)
        o, ov = F.conv2d(torch.relu(o), W2, b2, padding=1), cnv((o > 0).unsqueeze(1) * ov, W2, 1)
        h, hv = o + h, ov + hv
    W, b = sd[f"{p}module.{n+2}.weight"], sd[f"{p}module.{n+2}.bias"]
    o, ov = F.conv2d(torch.relu(h), W, b), cnv((h > 0).unsqueeze(1) * hv, W, 0)
    th = torch.tanh(o)
    w = sd[p + "weights"]
    return w * th + sd[p + "bias"], w * ((1 - th ** 2).unsqueeze(1) * ov)


# --------------------------------------------------------------------------------------
# affine coupling layers (acl.py) in the three mask flavours
# --------------------------------------------------------------------------------------


def checkerboard_mask(shape, reverse, like):
    """acl.py:68-78: mask[i, j] = (i + j) % 2 == 1, expanded over channels; 1 - mask if reversed."""
    C, H, W = shape
    ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    m = ((ii + jj) % 2 == 1).to(like.dtype).expand(C, H, W)
    return (1 - m) if reverse else m


def _cw_split(op, t, cdim=1):
    """acl.py:148-160 (+ :185-189 split-channel, :207-214 alternating): (passthrough, modified)."""
    C = op["x_shape"][0]
    idx = [slice(None)] * t.dim()
    if op["mask_type"] == "split-channel":
        k = C // 2
        a, b = idx.copy(), idx.copy()
        a[cdim], b[cdim] = slice(0, k), slice(k, None)
    else:
        a, b = idx.copy(), idx.copy()
        a[cdim], b[cdim] = slice(0, None, 2), slice(1, None, 2)
    first, second = t[tuple(a)], t[tuple(b)]
    return (second, first) if op["reverse"] else (first, second)


def _cw_combine(op, passthrough, modified, cdim=1):
    if op["reverse"]:
        passthrough, modified = modified, passthrough
    first, second = passthrough, modified
    if op["mask_type"] == "split-channel":
        return torch.cat((first, second), dim=cdim)
    shape = list(first.shape)
    shape[cdim] = first.shape[cdim] + second.shape[cdim]
    out = first.new_empty(shape)
    a, b = [slice(None)] * out.dim(), [slice(None)] * out.dim()
    a[cdim], b[cdim] = slice(0, None, 2), slice(1, None, 2)
    out[tuple(a)], out[tuple(b)] = first, second
    return out


def _chunk(y, cdim=1):
    """couplers.py:52-59 ChunkedSharedCoupler._split: first half shift, second half log-scale."""
    n = y.shape[cdim] // 2
    return y.narrow(cdim, 0, n), y.narrow(cdim, n, n)


def acl_x_to_z(sd, op, x):
    """acl.py:43-46 (checkerboard) / :101-111 (channelwise).  Returns (z, log-jac (B,1))."""
    if op["mask_type"] == "checkerboard":
        m = sd[op["prefix"] + "mask"]
        t, s = _chunk(net_forward(sd, op, m * x))
        z = m * x + (1 - m) * ((x + t) * torch.exp(s))
        return z, ((1 - m) * s).flatten(1).sum(1, keepdim=True)
    xp, xm = _cw_split(op, x)
    t, s = _chunk(net_forward(sd, op, xp))
    return _cw_combine(op, xp, (xm + t) * torch.exp(s)), s.flatten(1).sum(1, keepdim=True)


def acl_z_to_x(sd, op, z):
    """acl.py:48-51 / :113-123."""
    if op["mask_type"] == "checkerboard":
        m = sd[op["prefix"] + "mask"]
        t, s = _chunk(net_forward(sd, op, m * z))
        return m * z + (1 - m) * (z * torch.exp(-s) - t)
    zp, zm = _cw_split(op, z)
    t, s = _chunk(net_forward(sd, op, zp))
    return _cw_combine(op, zp, zm * torch.exp(-s) - t)


def acl_jvp(sd, op, z, v):
    """acl.py:53-66 / :125-146, one tangent per sample, primal recomputed (reference-equivalent)."""
    if op["mask_type"] == "checkerboard":
        m = sd[op["prefix"] + "mask"]
        y, yv = net_jvp(sd, op, m * z, m * v)
        (t, s), (tv, sv) = _chunk(y), _chunk(yv)
        x = m * z + (1 - m) * (z * torch.exp(-s) - t)
        jv = m * v + (1 - m) * (torch.exp(-s) * ((1 - m) * v - (1 - m) * z * sv) - tv)
        return x, jv
    (zp, zm), (vp, vm) = _cw_split(op, z), _cw_split(op, v)
    y, yv = net_jvp(sd, op, zp, vp)
    (t, s), (tv, sv) = _chunk(y), _chunk(yv)
    x = _cw_combine(op, zp, zm * torch.exp(-s) - t)
    return x, _cw_combine(op, vp, torch.exp(-s) * (vm - zm * sv) - tv)


def acl_jvp_multi(sd, op, z, V):
    """Same as acl_jvp for V of shape (B, d, ...): all columns at once."""
    if op["mask_type"] == "checkerboard":
        m = sd[op["prefix"] + "mask"]
        y, yv = net_jvp_multi(sd, op, m * z, m * V)
        (t, s), (tv, sv) = _chunk(y), _chunk(yv, 2)
        x = m * z + (1 - m) * (z * torch.exp(-s) - t)
        e = torch.exp(-s).unsqueeze(1)
        return x, m * V + (1 - m) * (e * ((1 - m) * V - ((1 - m) * z).unsqueeze(1) * sv) - tv)
    (zp, zm), (vp, vm) = _cw_split(op, z), _cw_split(op, V, 2)
    y, yv = net_jvp_multi(sd, op, zp, vp)
    (t, s), (tv, sv) = _chunk(y), _chunk(yv, 2)
    x = _cw_combine(op, zp, zm * torch.exp(-s) - t)
    e = torch.exp(-s).unsqueeze(1)
    return x, _cw_combine(op, vp, e * (vm - zm.unsqueeze(1) * sv) - tv, 2)


# --------------------------------------------------------------------------------------
# reshapes, split padding, tail
# --------------------------------------------------------------------------------------


def squeeze_x_to_z(x, f):
    """reshaping.py:89-101 Squeeze2dBijection._reshape_x (space-to-depth)."""
    B, C, H, W = x.shape
    return x.reshape(B, C, H // f, f, W // f, f).permute(0, 1, 3, 5, 2, 4).reshape(B, C * f * f, H // f, W // f)


def squeeze_z_to_x(z, f):
    """reshaping.py:103-114 Squeeze2dBijection._reshape_z.  Works with extra leading dims folded in."""
    lead = z.shape[:-3]
    C, H, W = z.shape[-3:]
    t = z.reshape(-1, C // (f * f), f, f, H, W).permute(0, 1, 4, 2, 5, 3)
    return t.reshape(*lead, C // (f * f), H * f, W * f)


def tail_gather(sd, op, x):
    """non_square.py:381-384: flatten, permute, keep the first d."""
    return x.flatten(1)[:, sd[op["prefix"] + "permutation"]][:, : op["d"]]


def tail_scatter(sd, op, z):
    """non_square.py:397-404 low_dim_to_masked; works for (..., d) with leading dims."""
    D = int(np.prod(op["x_shape"]))
    padded = z.new_zeros(*z.shape[:-1], D)
    padded[..., : op["d"]] = z
    return padded[..., sd[op["prefix"] + "inverse_permutation"]].reshape(*z.shape[:-1], *op["x_shape"])


def gaussian_log_prob(w):
    """gaussian.py:9-22 with mean 0, stddev 1 (factory.py:196-201)."""
    flat = w.flatten(1)
    return -0.5 * flat.shape[1] * math.log(2 * math.pi) - 0.5 * (flat ** 2).sum(1, keepdim=True)


LOGIT_EPS = 1e-7     # math.py:41-53



# --------------------------------------------------------------------------------------
# NSF prior: rand-channel-perm + LULinear + masked autoregressive rational-quadratic spline
#
# PARITY UNPINNED.  The reference builds these layers from jrmcornish/nsf @ 8e3fe75 (a fork of bayesiains/nsf,
# .gitmodules:1-3), which is NOT under /root/reference (gitmodules/nsf is an empty directory), so no reference output can
# be generated.  What follows restates the PUBLISHED algorithm -- Durkan, Bekasov, Murray, Papamakarios, "Neural Spline
# Flows", NeurIPS 2019, and the public nsf code base's nde/transforms/{lu.py, autoregressive.py, made.py,
# splines/rational_quadratic.py} -- anchored on the reference's call sites: bijections/nsf.py:86-113 (constructor
# arguments: tails='linear', num_blocks, use_residual_blocks=True, random_mask=False, relu, no batch norm),
# bijections/linear.py:12-34 (LULinear(identity_init=True)), bijections/reshaping.py:32-43, schemas.py:87-103,586-626
# (NUM_BINS 8, TAIL_BOUND 3, layer order).  The tests check the restatement's own invariants (inverse o forward = id,
# log-det = autograd Jacobian, autoregressive structure of the MADE masks), not reference vectors.
# --------------------------------------------------------------------------------------

NSF_KINDS = ("rand-channel-perm", "linear", "nsf-ar")
MIN_BIN_WIDTH = MIN_BIN_HEIGHT = MIN_DERIVATIVE = 1e-3       # nsf defaults, not overridden at bijections/nsf.py:100-113
LU_EPS = 1e-3                                                # LULinear(eps=1e-3)


def lu_linear_matrices(sd, prefix):
    """LULinear: W = L U with unit-lower L and U whose diagonal is softplus(unconstrained) + eps; logabsdet = sum log diag U."""
    lo, up, ud = sd[prefix + "linear.lower_entries"], sd[prefix + "linear.upper_entries"], sd[prefix + "linear.unconstrained_upper_diag"]
    n = ud.shape[0]
    il, iu = np.tril_indices(n, k=-1), np.triu_indices(n, k=1)
    L = torch.eye(n, dtype=ud.dtype)
    L[il[0], il[1]] = lo
    diag = F.softplus(ud) + LU_EPS
    U = torch.diag(diag)
    U[iu[0], iu[1]] = up
    return L, U, torch.log(diag).sum()


def made_degrees(features, hidden, blocks):
    """Degrees of the MADE units (nde/made.py, random_mask=False): inputs 1..D, hidden units arange % max(1, D-1) + min(1, D-1)
    in every hidden layer, outputs = input degrees tiled `multiplier` times (each feature's parameters contiguous)."""
    d_in = torch.arange(1, features + 1)
    max_, min_ = max(1, features - 1), min(1, features - 1)
    d_hid = torch.arange(hidden) % max_ + min_
    return d_in, d_hid


def made_masks(features, hidden, multiplier):
    d_in, d_hid = made_degrees(features, hidden, None)
    m_init = (d_hid[:, None] >= d_in[None, :]).float()           # hidden <- input
    m_hid = (d_hid[:, None] >= d_hid[None, :]).float()            # hidden <- hidden
    d_out = d_in.repeat_interleave(multiplier)
    m_out = (d_out[:, None] > d_hid[None, :]).float()             # output <- hidden: STRICT
    return m_init, m_hid, m_out


def made_forward(sd, op, x):
    """MADE with masked residual blocks (pre-activation relu, x + W1 relu(W0 relu(x))), nde/made.py."""
    p = op["prefix"] + "flow.autoregressive_net."
    D, H, K = x.shape[1], op["hidden"], 3 * op["bins"] - 1
    m_init, m_hid, m_out = made_masks(D, H, K)
    m_init, m_hid, m_out = m_init.to(x.dtype), m_hid.to(x.dtype), m_out.to(x.dtype)
    h = F.linear(x, sd[p + "initial_layer.weight"] * m_init, sd[p + "initial_layer.bias"])
    for b in range(op["blocks"]):
        q = p + f"blocks.{b}.linear_layers."
        t = F.linear(F.relu(h), sd[q + "0.weight"] * m_hid, sd[q + "0.bias"])
        t = F.linear(F.relu(t), sd[q + "1.weight"] * m_hid, sd[q + "1.bias"])
        h = h + t
    return F.linear(h, sd[p + "final_layer.weight"] * m_out, sd[p + "final_layer.bias"])


def rq_spline(x, params, hidden, bins, tail_bound, inverse=False):
    """Elementwise monotone rational-quadratic spline with linear tails (Durkan et al. 2019, eqs. 4-8; the public code's
    unconstrained_rational_quadratic_spline).  x: (B, D); params: (B, D, 3 bins - 1).  Returns (y, logabsdet (B, D))."""
    uw, uh, ud = params[..., :bins] / math.sqrt(hidden), params[..., bins:2 * bins] / math.sqrt(hidden), params[..., 2 * bins:]
    const = math.log(math.exp(1 - MIN_DERIVATIVE) - 1)
    ud = F.pad(ud, (1, 1), value=const)                          # boundary derivatives 1: matches the linear tails
    inside = (x >= -tail_bound) & (x <= tail_bound)
    xc = x.clamp(-tail_bound, tail_bound)
    lo, hi = -tail_bound, tail_bound

    def knots(u, minimum):
        w = minimum + (1 - minimum * bins) * F.softmax(u, dim=-1)
        c = F.pad(torch.cumsum(w, -1), (1, 0))
        c = (hi - lo) * c + lo
        c = torch.cat((torch.full_like(c[..., :1], lo), c[..., 1:-1], torch.full_like(c[..., :1], hi)), -1)
        return c, c[..., 1:] - c[..., :-1]

    cw, w = knots(uw, MIN_BIN_WIDTH)
    ch, h = knots(uh, MIN_BIN_HEIGHT)
    der = MIN_DERIVATIVE + F.softplus(ud)
    loc = (ch if inverse else cw).clone()
    loc[..., -1] += 1e-6
    idx = ((xc[..., None] >= loc).sum(-1) - 1).clamp(0, bins - 1)[..., None]
    g = lambda t: t.gather(-1, idx)[..., 0]
    icw, ibw, ich, ih = g(cw), g(w), g(ch), g(h)
    delta = h / w
    idelta, d0, d1 = g(delta), g(der), g(der[..., 1:])
    if inverse:
        yy = xc - ich
        a = yy * (d0 + d1 - 2 * idelta) + ih * (idelta - d0)
        b = ih * d0 - yy * (d0 + d1 - 2 * idelta)
        c = -idelta * yy
        root = (2 * c) / (-b - torch.sqrt(b * b - 4 * a * c))
        out = root * ibw + icw
        th = root
    else:
        th = (xc - icw) / ibw
    t1 = th * (1 - th)
    den = idelta + (d0 + d1 - 2 * idelta) * t1
    if not inverse:
        out = ich + ih * (idelta * th * th + d0 * t1) / den
    dnum = idelta * idelta * (d1 * th * th + 2 * idelta * t1 + d0 * (1 - th) * (1 - th))
    lad = torch.log(dnum) - 2 * torch.log(den)
    if inverse:
        lad = -lad
    return torch.where(inside, out, x), torch.where(inside, lad, torch.zeros_like(lad))


def nsf_x_to_z(sd, op, x):
    """x -> (z, log-jac (B, 1)) of one NSF-prior layer."""
    k, p = op["kind"], op["prefix"]
    if k == "rand-channel-perm":                                 # reshaping.py:39-40
        return x[:, sd[p + "permutation"]], x.new_zeros(x.shape[0], 1)
    if k == "linear":                                            # linear.py:23-28: y = L (U x) + b
        L, U, ld = lu_linear_matrices(sd, p)
        return F.linear(F.linear(x, U), L, sd[p + "linear.bias"]), ld.expand(x.shape[0], 1)
    params = made_forward(sd, op, x).view(x.shape[0], x.shape[1], -1)      # nsf.py:26-31 -> AutoregressiveTransform.forward
    z, lad = rq_spline(x, params, op["hidden"], op["bins"], op["tail_bound"])
    return z, lad.sum(1, keepdim=True)


def nsf_z_to_x(sd, op, z):
    """z -> x of one NSF-prior layer, AS THE REFERENCE'S WRAPPERS DO IT."""
    k, p = op["kind"], op["prefix"]
    if k == "rand-channel-perm":                                 # reshaping.py:42-43
        return z[:, sd[p + "inverse_permutation"]]
    if k == "linear":
        # linear.py:30-34 calls self.linear(z) -- the FORWARD map -- in _z_to_x.  Kept: samples drawn through an nsf prior
        # go through L U z + b here exactly as they do in the reference (the density path x -> z is unaffected).
        L, U, _ = lu_linear_matrices(sd, p)
        return F.linear(F.linear(z, U), L, sd[p + "linear.bias"])
    x = torch.zeros_like(z)                                      # AutoregressiveTransform.inverse: D passes
    for _ in range(z.shape[1]):
        params = made_forward(sd, op, x).view(z.shape[0], z.shape[1], -1)
        x, _ = rq_spline(z, params, op["hidden"], op["bins"], op["tail_bound"], inverse=True)
    return x


# --------------------------------------------------------------------------------------
# encode  x -> (z_low, low_dim_elbo)      non_square.py:65-66, exact.py:23-30, split.py:15-24
# --------------------------------------------------------------------------------------


def split_ops(ops):
    ih = next(i for i, o in enumerate(ops) if o["kind"] == "head")
    ib = next(i for i, o in enumerate(ops) if o["kind"] == "base")
    return ops[:ih], ops[ih], ops[ih + 1: ib], ops[ib], ops[ib + 1:]


def prehead(pre_ops, x, noise=None):
    """Dequantisation + the elementwise bijections in front of the head (wrapper.py:28-30,
    math.py:41-105).  Returns (y, summed log-jac (B,1)).  ``noise`` replaces rand_like."""
    lj = x.new_zeros(x.shape[0], 1)
    for op in pre_ops:
        k = op["kind"]
        if k == "dequant":
            x = x + (noise if noise is not None else torch.rand_like(x))
        elif k == "scalar-mult":
            x = op["value"] * x
            lj = lj + math.log(abs(op["value"])) * x[0].numel()
        elif k == "scalar-add":
            x = x + op["value"]
        elif k == "logit":
            xc = x.clamp(LOGIT_EPS, 1 - LOGIT_EPS)
            lj = lj + (-torch.log(xc) - torch.log(1 - xc)).flatten(1).sum(1, keepdim=True)
            x = torch.log(x) - torch.log(1 - x)
        else:
            raise ValueError(k)
    return x, lj


def encode(sd, flow_ops, base, prior_ops, x):
    """x (head input) -> z_low (B,d), low_dim_elbo (B,1), earliest latent (B,d)."""
    h = x
    for op in flow_ops:
        k = op["kind"]
        if k == "acl":
            h, _ = acl_x_to_z(sd, op, h)             # log-jac discarded: non_square.py:157-158,177
        elif k == "flatten":
            h = h.flatten(1)
        elif k == "squeeze":
            h = squeeze_x_to_z(h, op["factor"])
        elif k == "split":
            h = torch.chunk(h, 2, dim=1)[0]          # split.py:16-17 (density_2's elbo discarded)
        else:
            raise ValueError(k)
    z_low = tail_gather(sd, base, h)
    u, lj = z_low, z_low.new_zeros(z_low.shape[0], 1)
    for op in prior_ops:
        k = op["kind"]
        if k == "flatten":
            u = u.flatten(1)
        elif k == "acl":
            u, l = acl_x_to_z(sd, op, u)
            lj = lj + l
        elif k == "affine":                          # affine.py:24-38
            ls, sh = sd[op["prefix"] + "log_scale"], sd[op["prefix"] + "shift"]
            u = u * torch.exp(ls) + sh
            lj = lj + ls.sum()
        elif k in NSF_KINDS:
            u, l = nsf_x_to_z(sd, op, u)
            lj = lj + l
        elif k == "gaussian":
            lj = lj + gaussian_log_prob(u)
        else:
            raise ValueError(k)
    return z_low, lj, u


def nested_elbos(sd, ops, y):
    """The "elbo" entry at every level of the nested ``prior-dict`` the head returns (non_square.py:126-129), outermost
    first: BijectionDensity = deeper elbo + log-jac (exact.py:23-30), SplitDensity = density_1's + the Gaussian log-prob of
    the dropped half (split.py:15-24), tail = its prior's (non_square.py:381-395), Gaussian = log-prob (gaussian.py:65-74).
    ``y`` = the head's input.  Returns a list of (B, 1) tensors, one per level."""
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    h, contrib = y, []
    for op in flow_ops:
        k = op["kind"]
        if k == "acl":
            h, lj = acl_x_to_z(sd, op, h)
            contrib.append(lj)
        elif k == "flatten":
            h = h.flatten(1)
            contrib.append(torch.zeros_like(lj) if contrib else h.new_zeros(h.shape[0], 1))
        elif k == "squeeze":
            h = squeeze_x_to_z(h, op["factor"])
            contrib.append(h.new_zeros(h.shape[0], 1))
        elif k == "split":
            h, h2 = torch.chunk(h, 2, dim=1)
            contrib.append(gaussian_log_prob(h2))
    u = tail_gather(sd, base, h)
    contrib.append(u.new_zeros(u.shape[0], 1))             # the tail forwards its prior's elbo
    for op in prior_ops:
        k = op["kind"]
        if k == "flatten":
            u = u.flatten(1)
            contrib.append(u.new_zeros(u.shape[0], 1))
        elif k == "acl":
            u, lj = acl_x_to_z(sd, op, u)
            contrib.append(lj)
        elif k == "affine":
            ls, sh = sd[op["prefix"] + "log_scale"], sd[op["prefix"] + "shift"]
            u = u * torch.exp(ls) + sh
            contrib.append(ls.sum().expand(u.shape[0], 1))
        elif k in NSF_KINDS:
            u, lj = nsf_x_to_z(sd, op, u)
            contrib.append(lj)
        elif k == "gaussian":
            contrib.append(gaussian_log_prob(u))
    out, acc = [], 0
    for c in reversed(contrib):
        acc = acc + c
        out.append(acc)
    return out[::-1]


# --------------------------------------------------------------------------------------
# decode + Jacobian assembly
# --------------------------------------------------------------------------------------


def flow_forward(sd, flow_ops, base, z):
    """non_square.py:313-320: z_low -> x_hat through the z_to_x stack."""
    h = tail_scatter(sd, base, z)
    for op in reversed(flow_ops):
        k = op["kind"]
        if k == "acl":
            h = acl_z_to_x(sd, op, h)
        elif k == "flatten":
            h = h.reshape(h.shape[0], *op["x_shape"])
        elif k == "squeeze":
            h = squeeze_z_to_x(h, op["factor"])
        elif k == "split":
            h = torch.cat((h, torch.zeros_like(h)), dim=1)   # split.py:50-52 pad_inputs
    return h


def jvp_forward(sd, flow_ops, base, z, v):
    """non_square.py:322-329, one tangent per sample (reference-equivalent)."""
    h, hv = tail_scatter(sd, base, z), tail_scatter(sd, base, v)
    for op in reversed(flow_ops):
        k = op["kind"]
        if k == "acl":
            h, hv = acl_jvp(sd, op, h, hv)
        elif k == "flatten":
            h, hv = h.reshape(h.shape[0], *op["x_shape"]), hv.reshape(hv.shape[0], *op["x_shape"])
        elif k == "squeeze":
            h, hv = squeeze_z_to_x(h, op["factor"]), squeeze_z_to_x(hv, op["factor"])
        elif k == "split":
            h, hv = torch.cat((h, torch.zeros_like(h)), 1), torch.cat((hv, torch.zeros_like(hv)), 1)
    return h, hv


def jvp_forward_multi(sd, flow_ops, base, z, V):
    """All columns at once.  V: (B, S, d) seed tangents (S columns)."""
    h, hv = tail_scatter(sd, base, z), tail_scatter(sd, base, V)
    for op in reversed(flow_ops):
        k = op["kind"]
        if k == "acl":
            h, hv = acl_jvp_multi(sd, op, h, hv)
        elif k == "flatten":
            h, hv = h.reshape(h.shape[0], *op["x_shape"]), hv.reshape(*hv.shape[:2], *op["x_shape"])
        elif k == "squeeze":
            h, hv = squeeze_z_to_x(h, op["factor"]), squeeze_z_to_x(hv, op["factor"])
        elif k == "split":
            h, hv = torch.cat((h, torch.zeros_like(h)), 1), torch.cat((hv, torch.zeros_like(hv)), 2)
    return h, hv


def jtj_ref_equivalent(sd, flow_ops, base, z):
    """non_square.py:298-311: column loop, stack, bmm.  Returns (JtJ, x_hat, J (B,D,d))."""
    B, d = z.shape
    cols = []
    for i in range(d):
        v = torch.zeros_like(z)
        v[:, i] = 1
        xh, jv = jvp_forward(sd, flow_ops, base, z, v)
        cols.append(jv.flatten(1))
    J = torch.stack(cols, dim=2)
    return torch.bmm(J.transpose(1, 2), J), xh, J


def jtj_batched(sd, flow_ops, base, z):
    B, d = z.shape
    V = torch.eye(d, dtype=z.dtype).expand(B, d, d)
    xh, JV = jvp_forward_multi(sd, flow_ops, base, z, V)
    J = JV.flatten(2).transpose(1, 2)                   # (B, D, d)
    return torch.bmm(J.transpose(1, 2), J), xh, J


def cholesky_logdet(jtj):
    """non_square.py:262-296: whole-batch eps*I retry (eps 1e-6, x10), 2*sum(log diag L).
    The reference loop is unbounded; the oracle caps at MAX_ATTEMPTS=6 (dead constant there)."""
    eps, attempts = 1e-6, 1
    d = jtj.shape[1]
    eye = torch.eye(d, dtype=jtj.dtype).expand_as(jtj)
    while True:
        L, info = torch.linalg.cholesky_ex(jtj)
        if int(info.max()) == 0 and bool(torch.isfinite(L).all()):
            break
        if attempts >= 6:
            raise RuntimeError("JtJ not positive definite after 6 jitter attempts")
        jtj = jtj + eps * eye
        attempts += 1
        eps *= 10
    logdet = 2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(1, keepdim=True)
    return logdet, jtj, attempts


def metric_l1(jtj, diagonal):
    """non_square.py:87-100: sum_k |G_kk| or sum_{i != j} |G_ij|."""
    if diagonal:
        return torch.diagonal(jtj, dim1=-2, dim2=-1).abs().sum(1, keepdim=True)
    d = jtj.shape[1]
    off = jtj.masked_select(~torch.eye(d, dtype=torch.bool)).view(jtj.shape[0], d * (d - 1))
    return off.abs().sum(1, keepdim=True)


# --------------------------------------------------------------------------------------
# top level: elbo / ood / extract_latent / sample
# --------------------------------------------------------------------------------------


def elbo(sd, ops, x, add_reconstruction=True, add_diagonal_metric_reg=False, add_offdiagonal_metric_reg=False,
         likelihood_wt=1., metric_wt=1., ood=False, noise=None, flavour="batched", return_parts=False):
    """density.elbo(x, **kw)["elbo"] as the reference computes it: pre-head wrappers
    (exact.py:23-30) around NonSquareHeadDensity._elbo (non_square.py:64-129)."""
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    y, lj_pre = prehead(pre, x, noise)
    z_low, low_elbo, _ = encode(sd, flow_ops, base, prior_ops, y)
    parts = {"z_low": z_low, "low_dim_elbo": low_elbo, "prehead_logjac": lj_pre, "head_input": y}
    if not np.isclose(likelihood_wt, 0.):
        fn = jtj_ref_equivalent if flavour == "ref_equivalent" else jtj_batched
        jtj, xh, J = fn(sd, flow_ops, base, z_low)
        logdet, jtj, attempts = cholesky_logdet(jtj)
        likelihood = low_elbo - logdet / 2.
        if add_diagonal_metric_reg:
            l1 = metric_l1(jtj, True)
        elif add_offdiagonal_metric_reg:
            l1 = metric_l1(jtj, False)
        else:
            l1 = 0
        parts.update(jtj=jtj, logdet=logdet, J=J, attempts=attempts)
    else:
        l1, likelihood = 0, 0
        xh = flow_forward(sd, flow_ops, base, z_low)
    recon = ((xh - y).flatten(1) ** 2).sum(-1, keepdim=True) if add_reconstruction else 0
    parts.update(x_hat=xh, recon=recon, l1=l1, likelihood=likelihood)
    if ood:
        out = {"likelihood": likelihood, "reconstruction-error": recon}
    else:
        out = {"elbo": likelihood_wt * likelihood - head["regularization_param"] * recon - metric_wt * l1 + lj_pre}
    if return_parts:
        out["parts"] = parts
    return out


def extract_latent(sd, ops, x, earliest_latent=False, noise=None):
    """non_square.py:55-62 through the wrappers (exact.py:36-37, wrapper.py:24-25)."""
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    y, _ = prehead(pre, x, noise)
    z_low, _, u = encode(sd, flow_ops, base, prior_ops, y)
    return u if earliest_latent else z_low


def prior_inverse(sd, prior_ops, u):
    """exact.py:32-41 / affine.py:30-34: latent noise -> z_low through the low-dim flow."""
    for op in reversed(prior_ops):
        k = op["kind"]
        if k == "acl":
            u = acl_z_to_x(sd, op, u)
        elif k == "affine":
            u = (u - sd[op["prefix"] + "shift"]) * torch.exp(-sd[op["prefix"] + "log_scale"])
        elif k in NSF_KINDS:
            u = nsf_z_to_x(sd, op, u)
    return u


def fixed_sample(sd, ops, noise=None):
    """density.fixed_sample(noise): gaussian.py `_fixed_samples` when noise is None, then
    prior flow inverse, tail scatter (non_square.py:412-414), decode, pre-head inverses
    (math.py:45-46,83-84,100-101)."""
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    if noise is None:
        noise = sd[prior_ops[-1]["prefix"] + "_fixed_samples"]
    z_low = prior_inverse(sd, prior_ops, noise)
    x = flow_forward(sd, flow_ops, base, z_low)
    for op in reversed(pre):
        k = op["kind"]
        if k == "logit":
            x = torch.sigmoid(x)
        elif k == "scalar-add":
            x = x - op["value"]
        elif k == "scalar-mult":
            x = x / op["value"]
    return x


def jtj_matvec(sd, ops, z_low, eps):
    """non_square.py:190-201 J^T (J eps) for eps of shape (B, d, S); exact, via the explicit
    batched Jacobian (the oracle's stand-in for jvp + autograd vjp)."""
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    jtj, xh, J = jtj_batched(sd, flow_ops, base, z_low)
    return torch.bmm(jtj, eps), xh


def hutchinson_surrogate(sd, ops, z_low, eps, solve="exact"):
    """non_square.py:203-258 forward value: mean_s sum_k (A^-1 eps)_ks (A eps)_ks with
    A = J^T J.  The reference's solve is gpytorch.utils.linear_cg @ fc2053b (un-vendored,
    parity unpinned); the oracle substitutes the exact solve, for which the value is
    mean_s ||eps_s||^2 identically."""
    w, xh = jtj_matvec(sd, ops, z_low, eps)
    pre, head, flow_ops, base, prior_ops = split_ops(ops)
    jtj, _, _ = jtj_batched(sd, flow_ops, base, z_low)
    u = torch.linalg.solve(jtj, eps)
    return (u * w).sum(1, keepdim=True).mean(2), xh, w


def cg_documented(jtj, eps, max_iter, tol, min_iter=None):
    """The build's own CG stopping rule (cmf_amd/csrc/hutch_cg.hip), restated for checking the kernel:
    x0 = 0, unit-normalised right-hand sides, stop a sample after iteration k >= min_iter once the mean over its
    probes of the relative residual norm is < tol; at most max_iter.  (gpytorch's linear_cg @ fc2053b, which the
    reference calls at non_square.py:241-247, is un-vendored: parity unpinned.)"""
    B, d, S = eps.shape
    if min_iter is None:
        min_iter = min(10, max_iter - 1) + 1 if max_iter > 1 else 1
    nb = eps.norm(dim=1, keepdim=True)
    r = eps / nb
    x, p = torch.zeros_like(r), r.clone()
    rr = (r * r).sum(1, keepdim=True)
    active = torch.ones(B, dtype=torch.bool)
    iters = torch.zeros(B, dtype=torch.int32)
    for k in range(1, max_iter + 1):
        q = torch.bmm(jtj, p)
        alpha = rr / (p * q).sum(1, keepdim=True)
        a = active.view(B, 1, 1)
        x = torch.where(a, x + alpha * p, x)
        r_new = r - alpha * q
        rr_new = (r_new * r_new).sum(1, keepdim=True)
        p = torch.where(a, r_new + (rr_new / rr) * p, p)
        r, rr = torch.where(a, r_new, r), torch.where(a, rr_new, rr)
        iters[active] = k
        done = (k >= min_iter) & (rr.sqrt().mean(2).squeeze(1) < tol)
        active = active & ~done
        if not active.any():
            break
    return x * nb, iters
