"""Bijections on the hot path, with the reference's protocol ``x_to_z / z_to_x / jvp``
(``cmf/models/components/bijections/bijection.py:7-64``): dict results with keys ``z``/``x``, ``log-jac``
(B, 1) and ``jvp``, shape asserts on entry and exit.  Arithmetic is done by the HIP kernels
(``cmf_amd.engine``); each class only contributes its geometry as index maps.

  Checkerboard2dAffineCouplingBijection           acl.py:30-78
  SplitChannelwiseAffineCouplingBijection         acl.py:169-189
  AlternatingChannelwiseAffineCouplingBijection   acl.py:192-214
  ViewBijection / Squeeze2dBijection              reshaping.py:61-114
  AffineBijection                                 affine.py:10-38
  LogitBijection / ScalarMultiplication / ScalarAddition   math.py:41-105
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import engine as E

__all__ = [
    "RandomChannelwisePermutationBijection", "LULinearBijection", "AutoregressiveRationalQuadraticSplineBijection",
    "Bijection", "AffineCouplingBijection", "Checkerboard2dAffineCouplingBijection",
    "SplitChannelwiseAffineCouplingBijection", "AlternatingChannelwiseAffineCouplingBijection", "ViewBijection",
    "Squeeze2dBijection", "AffineBijection", "LogitBijection", "ScalarMultiplicationBijection",
    "ScalarAdditionBijection",
]


class _DeviceMaps:
    """int32 index maps, built once on the host and cached per device."""

    def __init__(self):
        self._host, self._dev = {}, {}

    def put(self, name, arr):
        self._host[name] = np.ascontiguousarray(arr, dtype=np.int32)

    def get(self, name, device):
        key = (name, str(device))
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self._host[name]).to(device)
        return self._dev[key]


class Bijection(nn.Module):
    def __init__(self, x_shape, z_shape):
        super().__init__()
        self.x_shape, self.z_shape = tuple(x_shape), tuple(z_shape)
        self._maps = _DeviceMaps()

    # protocol -------------------------------------------------------------------------------
    def forward(self, inputs, mode, **kwargs):
        if mode == "x-to-z":
            assert inputs.shape[1:] == self.x_shape
            out = self._x_to_z(inputs, **kwargs)
            assert out["z"].shape[1:] == self.z_shape
        elif mode == "z-to-x":
            assert inputs.shape[1:] == self.z_shape
            out = self._z_to_x(inputs, **kwargs)
            assert out["x"].shape[1:] == self.x_shape
        elif mode == "jvp":
            v = kwargs.pop("v")
            assert inputs.shape[1:] == self.z_shape and v.shape[1:] == self.z_shape
            out = self._jvp(inputs, v, **kwargs)
            assert out["x"].shape[1:] == self.x_shape and out["jvp"].shape[1:] == self.x_shape
        else:
            assert False, f"Invalid mode {mode}"
        return out

    def x_to_z(self, x, **kwargs):
        return self(x, "x-to-z", **kwargs)

    def z_to_x(self, z, **kwargs):
        return self(z, "z-to-x", **kwargs)

    def jvp(self, z, v, **kwargs):
        return self(z, "jvp", v=v, **kwargs)

    # single-column tangent helpers for the public jvp API -------------------------------------
    def _wrap_tangent(self, v, layout):
        B = v.shape[0]
        N = int(np.prod(v.shape[1:]))
        ident = torch.arange(N, dtype=torch.int32, device=v.device)
        eps = v.reshape(B, N, 1).contiguous()            # "probe" form: column 0 carries v
        return E.seed_tangent(B, N, 16, layout, ident, N, v.device, eps=eps)

    @staticmethod
    def _unwrap_tangent(T, shape):
        return T.to_dense(1)[:, :, 0].reshape(T.B, *shape).contiguous()


# --------------------------------------------------------------------------------------------------
# affine coupling layers
# --------------------------------------------------------------------------------------------------


class AffineCouplingBijection(Bijection):
    """z = (x + t(x_pass)) * exp(s(x_pass)) on the modified elements; pass-through elsewhere."""

    def __init__(self, x_shape, coupler):
        super().__init__(x_shape=x_shape, z_shape=x_shape)
        self.coupler = coupler
        self.geom = E.Geometry(x_shape)

    @property
    def net(self):
        return self.coupler.shift_log_scale_net

    def _set_maps(self, zi, ti, cmod):
        """zi: flat element ids (in z) of the modified elements; ti: element of the coupler output (2*cmod
        channels) holding the shift of each; its log-scale sits cmod channels further (couplers.py:52-59)."""
        HW = self.geom.HW
        self._maps.put("zi", zi)
        self._maps.put("ti", ti)
        self._maps.put("si", cmod * HW + np.asarray(ti))
        self.n_mod, self.cmod = len(zi), cmod

    def maps(self, device):
        return {"zi": self._maps.get("zi", device), "si": self._maps.get("si", device),
                "ti": self._maps.get("ti", device), "n": self.n_mod}

    def view(self, device):
        raise NotImplementedError

    @property
    def layout(self):
        return "panel" if self.geom.image else "fmajor"

    # engine-level steps (in place on z / T) ----------------------------------------------------
    def encode_(self, z, lj=None):
        view = self.view(z.device)
        if E.mlp_coupler_supported(self.net, view, None, 2 * self.cmod):      # the whole layer in one launch
            return E.mlp_coupler(self.net, z, None, view, self.maps(z.device), decode=False, lj=lj)
        y, _, _ = E.net_primal(self.net, z, view, need_acts=False)
        E.acl_primal(z, y, self.maps(z.device), decode=False, lj=lj)

    # Structural zeros (``zero_in=True``, decided by ``FlowProgram`` from the layer list alone, never from data): every element
    # this layer's network reads is known to be zero for the primal AND for every tangent column -- the channels
    # ``SplitDensity.pad_inputs`` appended (split.py:50-52) in front of a coupler whose pass-through half is exactly those
    # (acl.py:148-160,169-189 with reverse_mask).  Then the network's tangent output is identically zero, so no tangent network
    # runs (x-dot_mod = e^{-s} v_mod, bit for bit what the full computation yields), and its primal output is the same for every
    # sample (one sample group, read with stride 0).  The reference burns the cycles; 1 coupler in 10 of the image models.
    def pass_elements(self):
        """Flat element ids of z the coupler network reads (its "pass-through" input)."""
        raise NotImplementedError

    def decode_(self, z, T=None, lj=None, ncols=None, zero_in=False, seed_columns=None):
        view = self.view(z.device)
        if zero_in and self.net.kind == "resnet":
            y, g = E.net_primal_zero_input(self.net, view, z.shape[0], z.device)
            if T is not None:
                E.acl_tangent(T, None, z, y, g, self.maps(z.device))
            return E.acl_primal(z, y, self.maps(z.device), decode=True, lj=lj)
        if ((T is None or (ncols is not None and ncols <= 15 and lj is None))
                and E.mlp_coupler_supported(self.net, view, T, 2 * self.cmod)):
            return E.mlp_coupler(self.net, z, T, view, self.maps(z.device), decode=True, lj=lj, ncols=ncols)
        # the split-precision tangent pass reads relu' from bit masks written by the primal pass (engine.BitMask)
        want = False if T is None else ("bits" if E.cfg().tangent == "bf16x3" else True)
        y, g, acts = E.net_primal(self.net, z, view, need_acts=want)
        if T is not None and seed_columns is not None:
            # first layer of the decode sweep (FlowProgram._seed_columns): T holds the tail's one-hot seeds, and only the columns
            # seeded at a pass-through element have a non-zero network tangent -- the network runs on those, packed
            sc = seed_columns
            if sc["n"] == 0:                                         # no latent sits on a pass-through element: no network tangent at all
                E.acl_tangent(T, None, z, y, g, self.maps(z.device))
            else:
                Tc = E.seed_tangent(T.B, T.N, sc["nc"], T.layout, sc["col_of"], sc["n"], z.device)
                YTc = E.net_tangent(self.net, Tc, view, acts)
                YT = E.expand_columns(YTc, T.nc, sc["colmap"])
                YT.compact = getattr(YTc, "compact", False)
                self._acl_tangent(T, YT, z, y, g)
        elif T is not None:
            YT = E.net_tangent(self.net, T, view, acts)
            self._acl_tangent(T, YT, z, y, g)                       # uses z BEFORE the primal update
        E.acl_primal(z, y, self.maps(z.device), decode=True, lj=lj)

    def _acl_tangent(self, T, YT, z, y, g):
        if getattr(YT, "compact", False):
            # checkerboard tail (engine.net_tangent): the network's tangent exists at the modified pixels only, stored compactly;
            # (s, g) are read through the same compact index maps
            cm = self.compact_maps(z.device)
            yc, gc = E.gather_primal(y, cm["y_idx"], cm["y_n"]), E.gather_primal(g, cm["y_idx"], cm["y_n"])
            return E.acl_tangent(T, YT, z, yc, gc, cm)
        E.acl_tangent(T, YT, z, y, g, self.maps(z.device))

    # reverse sweep (J^T w): primal decode that keeps what the adjoint needs, then the adjoint step ------------
    def decode_ctx_(self, z, zero_in=False):
        view = self.view(z.device)
        zb = z.clone()
        if zero_in and self.net.kind == "resnet":
            y, g = E.net_primal_zero_input(self.net, view, z.shape[0], z.device)
            acts = "zero-input"
        else:
            y, g, acts = E.net_primal(self.net, z, view, need_acts=True)
        E.acl_primal(z, y, self.maps(z.device), decode=True)
        return zb, y, g, acts

    def decode_vjp_(self, Ct, ctx):
        """Adjoint of the tangent update of ``decode_`` on the cotangent stack ``Ct`` (in place)."""
        zb, y, g, acts = ctx
        dev = zb.device
        if isinstance(acts, str):                          # zero_in: the rows the network reads are dropped by the split's adjoint
            return E.acl_cotangent(Ct, None, zb, y, g, self.maps(dev))
        YC = E.Tangent(Ct.B, y[0].numel(), Ct.nc, self.layout, dev)
        YC.data.zero_()
        E.acl_cotangent(Ct, YC, zb, y, g, self.maps(dev))
        E.net_cotangent(self.net, YC, self.view(dev), acts, Ct)

    # training: decode keeping what the backward needs, and the backward of that step ------------
    def decode_train_(self, z, T, keep=True, nc_hint=None, zero_in=False):
        """``decode_`` on (z, T) in place; returns the context ``decode_backward_`` consumes: the layer input, the network's
        outputs and activations and, with a tangent stack (``T`` not None), every layer's input tangent, the modified tangent
        rows before the update and the network's raw tangent.  ``keep=False`` (recomputation): only the layer's inputs (z and a
        copy of T: 1/17 of the hidden tangents of a ResNet coupler) are kept and the context is rebuilt in the backward pass.
        ``zero_in``: no tangent network, nothing saved for one (the primal network still runs on the whole batch: its
        per-sample activations are what the primal weight gradients are accumulated from)."""
        zero_in = zero_in and self.net.kind == "resnet"
        if not keep and T is not None and not zero_in:
            ctx = ("recompute", z.clone(), E.Tangent(T.B, T.N, T.nc, T.layout, T.data.device, data=T.data[: T.B * T.N * T.nc].clone()))
            self.decode_(z, T)
            return ctx
        view, maps = self.view(z.device), self.maps(z.device)
        zb = z.clone()
        # ``nc_hint``: primal-only pass (T is None) whose activations a later tangent sweep with that many column slots reads
        y, g, acts = E.net_primal(self.net, z, view, need_acts=E.train_acts_mode(self.net, view, z.shape[0], T, nc=nc_hint))
        saved = V = YT = None
        if T is not None:
            V = E.modified_rows(T, maps)
            if zero_in:
                saved = "zero-input"
            else:
                saved = []
                YT = E.net_tangent(self.net, T, view, acts, save=saved)
            E.acl_tangent(T, YT, z, y, g, maps)
        E.acl_primal(z, y, maps, decode=True)
        return zb, y, g, acts, saved, V, YT

    def decode_tangent_from_ctx_(self, ctx, T, save=False, zero_in=False):
        """The tangent half of ``decode_train_`` on a context whose primal half is already there (``decode_train_(z, None)``):
        pushes ``T`` through the layer in place.  ``save=False``: nothing is kept (the d-column sweep that only feeds the Gram
        matrix); ``save=True``: returns the context ``decode_backward_`` consumes for THIS stack (the layer's primal state is
        shared, not recomputed) -- the low-rank Hutchinson backward runs both on one primal decode."""
        zb, y, g, acts = ctx[:4]
        view, maps = self.view(zb.device), self.maps(zb.device)
        V = E.modified_rows(T, maps) if save else None
        if zero_in and self.net.kind == "resnet":
            saved, YT = "zero-input", None
        else:
            saved = [] if save else None
            YT = E.net_tangent(self.net, T, view, acts, save=saved)
        self._acl_tangent(T, YT, zb, y, g)                   # zb: the layer input, i.e. z BEFORE the primal update
        return (zb, y, g, acts, saved, V, YT) if save else None

    def decode_backward_(self, Ct, dx, ctx, grads):
        """Backward of ``decode_train_``: ``Ct`` (cotangent of the tangent stack, or None) and ``dx`` (cotangent of the primal
        tensor) are updated in place from "after the layer" to "before the layer"; parameter gradients accumulate into ``grads``
        (dict parameter -> tensor).  Order matters: the cross terms read the cotangent of the UPDATED tangent rows."""
        if ctx[0] == "recompute":                          # rebuild this layer's state from its inputs (one more tangent sweep)
            ctx = self.decode_train_(ctx[1], ctx[2], keep=True)
        zb, y, g, acts, saved, V, YT = ctx
        dev = zb.device
        view, maps = self.view(dev), self.maps(dev)
        dy = torch.zeros_like(y)
        dg = None
        zero_in = isinstance(saved, str)                   # no tangent network ran: s-dot = t-dot = 0 (decode_train_(zero_in=True))
        if Ct is not None and zero_in:
            # only the log-scale sees the tangent update (d/ds of e^{-s} v); no reverse sweep, no tangent weight gradients: the
            # cotangent of the network's (zero) input tangent lands on rows the split's adjoint drops
            dg = torch.zeros_like(g) if g is not None else None
            E.acl_cross_terms(Ct, V, None, zb, y, g, maps, None, dy, dg)
            E.acl_cotangent(Ct, None, zb, y, g, maps)
        elif Ct is not None:
            dg = torch.zeros_like(g) if g is not None else None
            dz_ct = torch.zeros_like(zb)
            E.acl_cross_terms(Ct, V, YT, zb, y, g, maps, dz_ct, dy, dg)
            YC = E.Tangent(Ct.B, y[0].numel(), Ct.nc, self.layout, dev)
            YC.data.zero_()
            E.acl_cotangent(Ct, YC, zb, y, g, maps)
            cross = {}
            E.net_cotangent(self.net, YC, view, acts, Ct, saved=saved, grads=grads, cross=cross)
        E.acl_primal_backward(dx, zb, y, maps, dy, decode=True)
        if Ct is not None and not zero_in:
            dx += dz_ct
        if self.net.kind == "resnet":
            E.net_primal_backward(self.net, zb, view, acts, y, g, dy, dg, grads, dx)
        else:
            E.mlp_primal_backward(self.net, zb, view, acts, dy, grads, dx, dh_extra=cross if Ct is not None else None)

    def encode_train_(self, z, lj=None):
        """``encode_`` keeping the layer input, the network output and its activations for ``encode_backward_``."""
        view = self.view(z.device)
        xb = z.clone()
        y, g, acts = E.net_primal(self.net, z, view, need_acts=E.train_acts_mode(self.net, view, z.shape[0]))
        E.acl_primal(z, y, self.maps(z.device), decode=False, lj=lj)
        return xb, y, g, acts

    def encode_backward_(self, dz, ctx, grads, dlj=None):
        """Backward of ``encode_train_`` in place on ``dz`` (cotangent of the layer output -> of its input); ``dlj`` (B,) is the
        cotangent of the layer's log-jacobian where it is used (the low-dimensional prior flows)."""
        xb, y, g, acts = ctx
        dev = xb.device
        dy = torch.zeros_like(y)
        E.acl_primal_backward(dz, xb, y, self.maps(dev), dy, decode=False, dlj=dlj)
        if self.net.kind == "resnet":
            E.net_primal_backward(self.net, xb, self.view(dev), acts, y, g, dy, None, grads, dz)
        else:
            E.mlp_primal_backward(self.net, xb, self.view(dev), acts, dy, grads, dz)

    # protocol ------------------------------------------------------------------------------------
    def _x_to_z(self, x):
        E.require_gpu(x)
        z = x.detach().clone().contiguous()
        lj = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        self.encode_(z, lj)
        return {"z": z, "log-jac": lj.view(-1, 1)}

    def _z_to_x(self, z):
        E.require_gpu(z)
        x = z.detach().clone().contiguous()
        lj = torch.zeros(z.shape[0], dtype=torch.float32, device=z.device)
        self.decode_(x, None, lj)
        return {"x": x, "log-jac": lj.view(-1, 1)}

    def _jvp(self, z, v):
        E.require_gpu(z)
        x = z.detach().clone().contiguous()
        T = self._wrap_tangent(v.detach(), self.layout)
        self.decode_(x, T, ncols=1)
        return {"x": x, "jvp": self._unwrap_tangent(T, self.x_shape)}


class Checkerboard2dAffineCouplingBijection(AffineCouplingBijection):
    def __init__(self, x_shape, coupler, reverse_mask):
        super().__init__(x_shape=x_shape, coupler=coupler)
        C, H, W = x_shape
        ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        m = ((ii + jj) % 2 == 1).astype(np.float32)          # 1 = pass-through (acl.py:68-78)
        if reverse_mask:
            m = 1 - m
        mask = np.broadcast_to(m, (C, H, W)).copy()
        self.register_buffer("mask", torch.from_numpy(mask))
        zi = np.flatnonzero(mask.reshape(-1) == 0)
        self._set_maps(zi, zi, cmod=C)          # the net sees all C channels: its output is indexed like z
        self._pass_elements = np.flatnonzero(mask.reshape(-1) != 0)
        # compact form of the network's output (engine.net_tangent: last hidden conv and 1x1 conv at the modified pixels only):
        # pixel (row, col) of the (1 - mask) set sits at row * W/2 + col // 2
        self._live = None
        if W % 2 == 0:
            HW, HWc = H * W, H * W // 2
            livepix = np.flatnonzero(m.reshape(-1) == 0)                      # row-major = compact order (W/2 per row)
            assert len(livepix) == HWc and (livepix // W * (W // 2) + livepix % W // 2 == np.arange(HWc)).all()
            cidx = np.full(HW, -1)
            cidx[livepix] = np.arange(HWc)
            ch, px = zi // HW, zi % HW
            self._maps.put("c_ti", ch * HWc + cidx[px])
            self._maps.put("c_si", (C + ch) * HWc + cidx[px])
            self._maps.put("c_y_idx", (np.arange(2 * C)[:, None] * HW + livepix[None, :]).reshape(-1))
            self._live = 2 if reverse_mask else 1                            # the modified pixels' (row + col) % 2 + 1
            self._livepix = livepix

    def view(self, device):
        live = None
        if self._live is not None and self.net.kind == "resnet":
            hid = self.net.module[0].out_channels
            key = f"c_act_idx_{hid}"
            if key not in self._maps._host:
                self._maps.put(key, (np.arange(hid)[:, None] * self.geom.HW + self._livepix[None, :]).reshape(-1))
            live = {"parity": self._live, "act_idx": self._maps.get(key, device)}
        return E.NetView(self.geom, cin=self.geom.C, mask=self.mask, live=live)

    def compact_maps(self, device):
        return {"zi": self._maps.get("zi", device), "si": self._maps.get("c_si", device), "ti": self._maps.get("c_ti", device),
                "n": self.n_mod, "y_idx": self._maps.get("c_y_idx", device), "y_n": 2 * self.geom.C * (self.geom.HW // 2)}

    def pass_elements(self):
        return self._pass_elements                           # the net reads mask . z (acl.py:48-52)


class _ChannelwiseACL(AffineCouplingBijection):
    def __init__(self, x_shape, coupler_factory, pass_idx, mod_idx):
        coupler = coupler_factory(len(pass_idx))
        super().__init__(x_shape=x_shape, coupler=coupler)
        HW = self.geom.HW
        zi = (np.asarray(mod_idx)[:, None] * HW + np.arange(HW)[None, :]).reshape(-1)
        self._set_maps(zi, np.arange(len(zi)), cmod=len(mod_idx))   # net output channel j <-> j-th modified channel
        self._pass = (int(pass_idx[0]), int(pass_idx[1] - pass_idx[0]) if len(pass_idx) > 1 else 1, len(pass_idx))

    def view(self, device):
        off, step, n = self._pass
        return E.NetView(self.geom, cin=n, chan_off=off, chan_step=step)

    def pass_elements(self):
        off, step, n = self._pass
        HW = self.geom.HW
        return ((off + step * np.arange(n))[:, None] * HW + np.arange(HW)[None, :]).reshape(-1)


class SplitChannelwiseAffineCouplingBijection(_ChannelwiseACL):
    """First half of the channels passes through (second half if reverse_mask): acl.py:169-189."""

    def __init__(self, x_shape, coupler_factory, reverse_mask):
        C = x_shape[0]
        first, second = np.arange(0, C // 2), np.arange(C // 2, C)
        p, m = (second, first) if reverse_mask else (first, second)
        assert len(p) > 0, "Not a bijection without passthrough"
        super().__init__(x_shape, coupler_factory, p, m)
        self.reverse_mask = reverse_mask


class AlternatingChannelwiseAffineCouplingBijection(_ChannelwiseACL):
    """Even channels pass through (odd if reverse_mask): acl.py:192-214."""

    def __init__(self, x_shape, coupler_factory, reverse_mask):
        C = x_shape[0]
        first, second = np.arange(0, C, 2), np.arange(1, C, 2)
        p, m = (second, first) if reverse_mask else (first, second)
        assert len(p) > 0, "Not a bijection without passthrough"
        super().__init__(x_shape, coupler_factory, p, m)
        self.reverse_mask = reverse_mask


# --------------------------------------------------------------------------------------------------
# reshapes
# --------------------------------------------------------------------------------------------------


class _ReshapingBijection(Bijection):
    """Pure index maps, applied identically to values and tangents (reshaping.py:8-29)."""

    def _index_maps(self):
        n = int(np.prod(self.x_shape))
        ids = torch.arange(n).reshape(1, *self.x_shape)
        self._maps.put("x2z", self._reshape_x(ids).reshape(-1).numpy())    # z[r] = x[x2z[r]]
        idz = torch.arange(n).reshape(1, *self.z_shape)
        self._maps.put("z2x", self._reshape_z(idz).reshape(-1).numpy())    # x[r] = z[z2x[r]]
        self.n = n

    def encode(self, x):
        return E.gather_primal(x, self._maps.get("x2z", x.device), self.n).view(x.shape[0], *self.z_shape)

    def decode(self, z, T=None):
        x = E.gather_primal(z, self._maps.get("z2x", z.device), self.n).view(z.shape[0], *self.x_shape)
        if T is not None:
            T = E.gather_tangent(T, self._maps.get("z2x", z.device), self.n)
        return x, T

    def decode_tangent(self, T):
        """``decode`` on a tangent stack alone (the primal tensor went through it in an earlier pass)."""
        return E.gather_tangent(T, self._maps.get("z2x", T.data.device), self.n)

    def decode_vjp(self, Ct):
        """Adjoint of ``decode`` on a cotangent stack: the inverse index map."""
        return E.gather_tangent(Ct, self._maps.get("x2z", Ct.data.device), self.n)

    def _zeros(self, t):
        return torch.zeros(t.shape[0], 1, dtype=t.dtype, device=t.device)

    def _x_to_z(self, x):
        E.require_gpu(x)
        return {"z": self.encode(x.contiguous()), "log-jac": self._zeros(x)}

    def _z_to_x(self, z):
        E.require_gpu(z)
        return {"x": self.decode(z.contiguous())[0], "log-jac": self._zeros(z)}

    def _jvp(self, z, v):
        E.require_gpu(z)
        return {"x": self.decode(z.contiguous())[0], "jvp": self.decode(v.contiguous())[0]}


class ViewBijection(_ReshapingBijection):
    def __init__(self, x_shape, z_shape):
        assert np.prod(x_shape) == np.prod(z_shape)
        super().__init__(x_shape, z_shape)
        self.n = int(np.prod(x_shape))

    # a contiguous view: no data movement at all
    def encode(self, x):
        return x.view(x.shape[0], *self.z_shape)

    def decode(self, z, T=None):
        return z.view(z.shape[0], *self.x_shape), T

    def decode_tangent(self, T):
        return T

    def decode_vjp(self, Ct):
        return Ct


class Squeeze2dBijection(_ReshapingBijection):
    """Space-to-depth by ``factor`` (reshaping.py:69-114)."""

    def __init__(self, x_shape, factor):
        C, H, W = x_shape
        assert H % factor == 0 and W % factor == 0
        self.factor = factor
        super().__init__(x_shape, (C * factor * factor, H // factor, W // factor))
        self._index_maps()

    def _reshape_x(self, x):
        f, (C, H, W) = self.factor, self.x_shape
        return x.reshape(-1, C, H // f, f, W // f, f).permute(0, 1, 3, 5, 2, 4).reshape(-1, *self.z_shape)

    def _reshape_z(self, z):
        f, (C, H, W) = self.factor, self.x_shape
        return z.reshape(-1, C, f, f, H // f, W // f).permute(0, 1, 4, 2, 5, 3).reshape(-1, *self.x_shape)


# --------------------------------------------------------------------------------------------------
# NSF prior layers (schemas.py:87-103,586-626): PARITY UNPINNED -- jrmcornish/nsf @ 8e3fe75 is not vendored
# under the reference tree; built from the published algorithm (csrc/nsf.hip), tested against oracle/cmf_oracle.py's
# restatement.  Module / parameter names follow the nsf code base so that the state-dict keys are the reference's:
#   bijection.{permutation, inverse_permutation}
#   bijection.linear.{bias, lower_entries, upper_entries, unconstrained_upper_diag}
#   bijection.flow.autoregressive_net.{initial_layer, blocks.i.linear_layers.j, final_layer}.{weight, bias, mask, degrees}
# These layers sit BELOW the tail: they act on (B, d) latents in the encode / sampling directions only, no tangents.
# --------------------------------------------------------------------------------------------------


class _PriorFlowLayer(Bijection):
    """A low-dimensional prior layer evaluated out of place: ``prior_encode(u, lj) -> z`` (lj (B,) accumulates the
    log-jacobian) and ``prior_decode(z) -> u`` as the reference's ``z_to_x`` computes it."""

    def prior_encode(self, u, lj=None):
        raise NotImplementedError

    def prior_decode(self, z):
        raise NotImplementedError

    def prior_encode_train(self, u, lj=None):
        """``prior_encode`` keeping what ``prior_backward`` needs: returns (z, ctx)."""
        raise NotImplementedError

    def prior_backward(self, dz, ctx, grads, dlj=None):
        """Cotangent of the layer input from ``dz`` (B, d) and the cotangent ``dlj`` (B,) of the layer's log-jacobian;
        parameter gradients accumulate into ``grads``."""
        raise NotImplementedError

    def _x_to_z(self, x):
        E.require_gpu(x)
        lj = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        return {"z": self.prior_encode(x.detach().contiguous(), lj), "log-jac": lj.view(-1, 1)}

    def _z_to_x(self, z):
        # the wrappers return a log-jac too; the hot path never reads it in this direction (sampling)
        E.require_gpu(z)
        x = self.prior_decode(z.detach().contiguous())
        return {"x": x, "log-jac": -self._x_to_z(x)["log-jac"]}


class RandomChannelwisePermutationBijection(_PriorFlowLayer):
    """z = x[:, permutation] (reshaping.py:32-43); the permutation is drawn at construction and saved in checkpoints."""

    def __init__(self, x_shape):
        assert len(x_shape) == 1
        super().__init__(x_shape, x_shape)
        self.register_buffer("permutation", torch.randperm(x_shape[0]))
        self.register_buffer("inverse_permutation", torch.argsort(self.permutation))

    def prior_encode(self, u, lj=None):
        return E.gather_primal(u, self.permutation.to(torch.int32), u.shape[1])

    def prior_decode(self, z):
        return E.gather_primal(z, self.inverse_permutation.to(torch.int32), z.shape[1])

    def prior_encode_train(self, u, lj=None):
        return self.prior_encode(u, lj), None

    def prior_backward(self, dz, ctx, grads, dlj=None):        # z[j] = x[perm[j]]  ->  dx[i] = dz[inverse_perm[i]]
        return E.gather_primal(dz.contiguous(), self.inverse_permutation.to(torch.int32), dz.shape[1])


class _LULinearParameters(nn.Module):
    """Parameter container with the names (and registration order) of nsf's ``LULinear(features, identity_init=True)``."""

    def __init__(self, features, eps=1e-3):
        super().__init__()
        self.features, self.eps = features, eps
        n_tri = (features - 1) * features // 2
        self.bias = nn.Parameter(torch.zeros(features))
        self.lower_entries = nn.Parameter(torch.zeros(n_tri))
        self.upper_entries = nn.Parameter(torch.zeros(n_tri))
        self.unconstrained_upper_diag = nn.Parameter(torch.full((features,), float(np.log(np.exp(1 - eps) - 1))))


class LULinearBijection(_PriorFlowLayer):
    """z = L (U x) + b, log-jac = sum log diag(U) (bijections/linear.py:12-28 -> nsf LULinear)."""

    def __init__(self, num_input_channels):
        super().__init__((num_input_channels,), (num_input_channels,))
        self.linear = _LULinearParameters(num_input_channels)

    def _affine_map(self, u, lj):
        p = self.linear
        W, ld = E.lu_weights(p.lower_entries, p.upper_entries, p.unconstrained_upper_diag, p.eps)
        if lj is not None:
            lj += ld                                   # the same constant for every sample
        return E.linear_primal(u, W, p.bias.detach())

    def prior_encode(self, u, lj=None):
        return self._affine_map(u, lj)

    def prior_decode(self, z):
        # bijections/linear.py:30-34: ``_z_to_x`` calls ``self.linear(z)`` -- the FORWARD map, not the inverse.  Kept as is:
        # samples drawn through an nsf prior see L U z + b here exactly as in the reference (the density path is unaffected).
        return self._affine_map(z, None)

    def prior_encode_train(self, u, lj=None):
        return self._affine_map(u, lj), u

    def prior_backward(self, dz, ctx, grads, dlj=None):
        """z = L U x + b, log-jac = sum log diag U: dW = sum dz (x) x through the weight-gradient kernel, then the triangles'
        and the diagonal's gradients (cmf_lu_backward); dx = (L U)^T dz."""
        p, n = self.linear, dz.shape[1]
        W, _ = E.lu_weights(p.lower_entries, p.upper_entries, p.unconstrained_upper_diag, p.eps)
        gb = E.GroupedBatch(dz.shape[0], dz.device)
        dW = torch.zeros(n, n, dtype=torch.float32, device=dz.device)
        dx_g = gb.linear_backward(gb.pack(ctx), gb.pack(dz), W, n, n, dw=dW, db=E._grad_of(grads, p.bias))
        scratch = torch.empty(1, dtype=torch.float32, device=dz.device)
        g = lambda t: E._p(E._grad_of(grads, t))
        E._lib.check(E._lib.load().cmf_lu_backward(
            E._p(dW), E._p(p.lower_entries.detach().contiguous()), E._p(p.upper_entries.detach().contiguous()),
            E._p(p.unconstrained_upper_diag.detach().contiguous()), n, float(p.eps),
            E._p(None if dlj is None else dlj.to(torch.float32).contiguous()), dz.shape[0], E._p(scratch),
            g(p.lower_entries), g(p.upper_entries), g(p.unconstrained_upper_diag), E._stream()), "cmf_lu_backward")
        return gb.unpack(dx_g, n).contiguous()


class _MaskedLinear(nn.Linear):
    """nsf's MaskedLinear: an nn.Linear with ``mask`` / ``degrees`` buffers (random_mask = False)."""

    def __init__(self, in_degrees, out_features, autoregressive_features, is_output):
        super().__init__(len(in_degrees), out_features, bias=True)
        if is_output:
            out_degrees = torch.arange(1, autoregressive_features + 1).repeat_interleave(out_features // autoregressive_features)
            mask = (out_degrees[:, None] > in_degrees[None, :]).float()
        else:
            max_, min_ = max(1, autoregressive_features - 1), min(1, autoregressive_features - 1)
            out_degrees = torch.arange(out_features) % max_ + min_
            mask = (out_degrees[:, None] >= in_degrees[None, :]).float()
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", out_degrees)
        self.kind = 2 if is_output else None           # 0 / 1 set by the owner (input or hidden source)

    forward = None


class _MaskedResidualBlock(nn.Module):
    def __init__(self, in_degrees, autoregressive_features):
        super().__init__()
        features = len(in_degrees)
        l0 = _MaskedLinear(in_degrees, features, autoregressive_features, False)
        l1 = _MaskedLinear(l0.degrees, features, autoregressive_features, False)
        l0.kind = l1.kind = 1
        self.linear_layers = nn.ModuleList([l0, l1])
        self.degrees = l1.degrees
        nn.init.uniform_(l1.weight, -1e-3, 1e-3)       # zero_initialization=True
        nn.init.uniform_(l1.bias, -1e-3, 1e-3)


class _MADE(nn.Module):
    def __init__(self, features, hidden_features, num_blocks, output_multiplier):
        super().__init__()
        self.features, self.hidden_features, self.multiplier = features, hidden_features, output_multiplier
        self.initial_layer = _MaskedLinear(torch.arange(1, features + 1), hidden_features, features, False)
        self.initial_layer.kind = 0
        deg = self.initial_layer.degrees
        blocks = []
        for _ in range(num_blocks):
            blocks.append(_MaskedResidualBlock(deg, features))
            deg = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = _MaskedLinear(deg, features * output_multiplier, features, True)

    def _w(self, lin):
        return E.made_masked_weight(lin.weight, lin.kind, self.features, self.multiplier)

    def evaluate(self, x, save=None):
        """(B, D) -> (B, D * multiplier) spline parameters: masked linear layers through cmf_conv_primal (taps = 1), relu on
        load, residual add in the epilogue.  ``save`` (a list, training): the input of every block and its inner activation."""
        w = self._w
        h = E.linear_primal(x, w(self.initial_layer), self.initial_layer.bias.detach())
        for blk in self.blocks:
            l0, l1 = blk.linear_layers
            t = E.linear_primal(h, w(l0), l0.bias.detach(), relu_in=True)
            if save is not None:
                save.append((h, t))
            h = E.linear_primal(t, w(l1), l1.bias.detach(), relu_in=True, res=h)
        if save is not None:
            save.append(h)
        return E.linear_primal(h, w(self.final_layer), self.final_layer.bias.detach())

    def backward(self, x, saved, dtheta, grads):
        """Backward of ``evaluate``: parameter gradients (masked like the weights) into ``grads``, returns d x (B, D).
        Residual block  h' = h + W1 relu(W0 relu(h) + b0) + b1  on the tangent-conv kernels with 16 samples in the column
        slots: weight gradients with the input's own relu, transposed products with the relu' output factor."""
        B, dev = x.shape[0], x.device
        gb = E.GroupedBatch(B, dev)
        H_, D, K = self.hidden_features, self.features, self.multiplier
        lib = E._lib.load()

        def wgrad_masked(lin):
            g = E._grad_of(grads, lin.weight)
            return g, lambda: E._lib.check(lib.cmf_made_mask_weight(E._p(g), E._p(g), g.shape[0], g.shape[1], lin.kind, D, K,
                                                                      E._stream()), "cmf_made_mask_weight")

        fl = self.final_layer
        g, mask = wgrad_masked(fl)
        dh = gb.linear_backward(gb.pack(saved[-1]), gb.pack(dtheta), self._w(fl), H_, D * K, dw=g, db=E._grad_of(grads, fl.bias))
        mask()
        for blk, (h_in, t) in zip(reversed(list(self.blocks)), reversed(saved[:-1])):
            l0, l1 = blk.linear_layers
            h_g, t_g = gb.pack(h_in), gb.pack(t)
            g, mask = wgrad_masked(l1)
            dt = gb.linear_backward(t_g, dh, self._w(l1), H_, H_, dw=g, db=E._grad_of(grads, l1.bias), relu_in=True, fo_g=t_g)
            mask()
            g, mask = wgrad_masked(l0)
            dh = gb.linear_backward(h_g, dt, self._w(l0), H_, H_, dw=g, db=E._grad_of(grads, l0.bias), relu_in=True, fo_g=h_g,
                                    res_g=dh)
            mask()
        il = self.initial_layer
        g, mask = wgrad_masked(il)
        dx_g = gb.linear_backward(gb.pack(x), dh, self._w(il), D, H_, dw=g, db=E._grad_of(grads, il.bias))
        mask()
        return gb.unpack(dx_g, D)


class _AutoregressiveSplineFlow(nn.Module):
    def __init__(self, features, hidden_features, num_blocks, num_bins, tail_bound):
        super().__init__()
        self.num_bins, self.tail_bound = num_bins, tail_bound
        self.autoregressive_net = _MADE(features, hidden_features, num_blocks, 3 * num_bins - 1)


class AutoregressiveRationalQuadraticSplineBijection(_PriorFlowLayer):
    """Masked autoregressive rational-quadratic spline (bijections/nsf.py:86-113 -> nsf
    MaskedPiecewiseRationalQuadraticAutoregressiveTransform with tails = 'linear', residual MADE, relu, no dropout)."""

    def __init__(self, num_input_channels, num_hidden_layers, num_hidden_channels, num_bins, tail_bound):
        super().__init__((num_input_channels,), (num_input_channels,))
        self.flow = _AutoregressiveSplineFlow(num_input_channels, num_hidden_channels, num_hidden_layers, num_bins, tail_bound)

    def prior_encode(self, u, lj=None):
        f = self.flow
        params = f.autoregressive_net.evaluate(u)
        return E.rq_spline(u, params, f.num_bins, f.autoregressive_net.hidden_features, f.tail_bound, inverse=False, lj=lj)

    def prior_encode_train(self, u, lj=None):
        f = self.flow
        saved = []
        params = f.autoregressive_net.evaluate(u, save=saved)
        z = E.rq_spline(u, params, f.num_bins, f.autoregressive_net.hidden_features, f.tail_bound, inverse=False, lj=lj)
        return z, (u, params, saved)

    def prior_backward(self, dz, ctx, grads, dlj=None):
        """z_f = spline(x_f; theta_f(x_{<f})): the direct term dz dz/dx + dlj dlad/dx and the MADE's backward of the parameter
        cotangents dz dz/dtheta + dlj dlad/dtheta (cmf_rq_spline_backward: forward-mode duals)."""
        f, (u, params, saved) = self.flow, ctx
        net = f.autoregressive_net
        dx, dparams = E.rq_spline_backward(u, params, f.num_bins, net.hidden_features, f.tail_bound, dz,
                                           None if dlj is None else dlj.to(torch.float32).contiguous())
        dx_net = net.backward(u, saved, dparams, grads)
        E.accumulate_any(dx, dx_net.contiguous())
        return dx

    def prior_decode(self, z):
        """AutoregressiveTransform.inverse: D passes of (MADE, elementwise inverse spline); pass i fixes feature i."""
        f = self.flow
        x = torch.zeros_like(z)
        for _ in range(z.shape[1]):
            params = f.autoregressive_net.evaluate(x)
            x = E.rq_spline(z, params, f.num_bins, f.autoregressive_net.hidden_features, f.tail_bound, inverse=True)
        return x


# --------------------------------------------------------------------------------------------------
# 2-D prior and pre-head elementwise maps
# --------------------------------------------------------------------------------------------------


class AffineBijection(Bijection):
    """z = x * exp(log_scale) + shift, per element (affine.py:10-38, per_channel=False)."""

    def __init__(self, x_shape, per_channel=False):
        assert not per_channel, "only the per-element form is on the non-square path (schemas.py:75-79)"
        super().__init__(x_shape, x_shape)
        self.shift = nn.Parameter(torch.zeros(x_shape))
        self.log_scale = nn.Parameter(torch.zeros(x_shape))

    def encode_(self, z, lj=None):
        E.affine_prior(z, self.log_scale, self.shift, decode=False, lj=lj)

    def decode_(self, z, lj=None):
        E.affine_prior(z, self.log_scale, self.shift, decode=True, lj=lj)

    def encode_train_(self, z, lj=None):
        xb = z.clone()
        self.encode_(z, lj)
        return xb

    def encode_backward_(self, dz, ctx, grads, dlj=None):
        """u = x e^{ls} + sh, lj = sum ls: parameter gradients and dz <- dz e^{ls} in one kernel (cmf_affine_prior_backward)."""
        B = dz.shape[0]
        d2, x2 = dz.view(B, -1), ctx.view(B, -1)
        n = d2.shape[1]
        E._lib.check(E._lib.load().cmf_affine_prior_backward(
            E._p(d2), n, E._p(x2), n, E._p(self.log_scale.detach().contiguous()), n, B,
            E._p(None if dlj is None else dlj.to(torch.float32).contiguous()), E._p(E._grad_of(grads, self.log_scale).view(-1)),
            E._p(E._grad_of(grads, self.shift).view(-1)), E._stream()), "cmf_affine_prior_backward")

    def _x_to_z(self, x):
        E.require_gpu(x)
        z, lj = x.detach().clone().contiguous(), torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        self.encode_(z, lj)
        return {"z": z, "log-jac": lj.view(-1, 1)}

    def _z_to_x(self, z):
        E.require_gpu(z)
        x, lj = z.detach().clone().contiguous(), torch.zeros(z.shape[0], dtype=torch.float32, device=z.device)
        self.decode_(x, lj)
        return {"x": x, "log-jac": lj.view(-1, 1)}


class _Elementwise(Bijection):
    """y = a*x + c (optionally followed by logit): the pre-head chain of image configs (math.py:9-105).
    A run of these in front of the head is fused into one kernel by ``FlowProgram``."""
    a, c, logit = 1.0, 0.0, False

    def __init__(self, x_shape):
        super().__init__(x_shape, x_shape)

    def _x_to_z(self, x):
        E.require_gpu(x)
        z, lj = E.prehead(x.contiguous(), None, self.a, self.c, self.logit)
        return {"z": z, "log-jac": lj.view(-1, 1)}

    def _z_to_x(self, z):
        E.require_gpu(z)
        x = E.prehead_inverse(z.contiguous(), self.a, self.c, self.logit)
        lj = -self._x_to_z(x)["log-jac"]
        return {"x": x, "log-jac": lj}


class LogitBijection(_Elementwise):
    logit = True


class ScalarMultiplicationBijection(_Elementwise):
    def __init__(self, x_shape, value):
        assert np.isscalar(value) and value != 0., "Scalar multiplication by zero is not a bijection"
        super().__init__(x_shape)
        self.value = self.a = float(value)


class ScalarAdditionBijection(_Elementwise):
    def __init__(self, x_shape, value):
        assert np.isscalar(value)
        super().__init__(x_shape)
        self.value = self.c = float(value)
