"""Density modules with the reference's API (``cmf/models/components/densities``): mode-dispatch
``forward`` and ``elbo / sample / fixed_sample / jvp / ood / extract_latent`` with the same kwargs
and result-dict keys, the same nesting (``.prior`` / ``.density`` / ``.density_1`` / ``.module``)
and therefore the same state-dict key schema.

What differs is how the non-square head evaluates its log-density.  The reference's
``NonSquareHeadDensity._elbo`` (non_square.py:64-129) re-walks the module chain on every call,
builds two stacks of bound methods (:146-188) and pushes ONE Jacobian column per decode of the whole
stack (:298-311).  Here the chain is compiled once into a ``FlowProgram`` that keeps the primal in
the reference's (B, C, H, W) layout and carries ALL d Jacobian columns of a sample side by side
(column-innermost tangent tensors), so each coupler-network layer is a single MFMA kernel launch
for the whole batch and all columns, followed by the fused Gram + Cholesky + log-det kernel.
"""
import numpy as np
import torch
import torch.nn as nn

from . import engine as E
from .bijections import AffineBijection, AffineCouplingBijection, _Elementwise, _PriorFlowLayer, _ReshapingBijection

__all__ = ["Density", "BijectionDensity", "SplitDensity", "DiagonalGaussianDensity", "WrapperDensity",
           "DequantizationDensity", "DataParallelDensity", "NonSquareHeadDensity", "ManifoldFlowHeadDensity",
           "NonSquareTailDensity", "FlowProgram"]


class Density(nn.Module):
    """Mode-string dispatch kept from the reference (density.py:7-28)."""

    def forward(self, mode, *args, **kwargs):
        if mode == "elbo":
            return self._elbo(*args, **kwargs)
        if mode == "sample":
            return self._sample(*args)
        if mode == "fixed-sample":
            return self._fixed_sample(*args)
        if mode == "jvp":
            return self._jvp(*args)
        if mode == "ood":
            return self._ood(*args)
        if mode == "extract-latent":
            return self._extract_latent(*args, **kwargs)
        assert False, f"Invalid mode {mode}"

    def elbo(self, x, **kwargs):
        return self("elbo", x, **kwargs)

    def sample(self, num_samples):
        return self("sample", num_samples)

    def fixed_sample(self, noise=None):
        return self("fixed-sample", noise)

    def jvp(self, x, v):
        return self("jvp", x, v)

    def ood(self, x):
        return self("ood", x)

    def extract_latent(self, x, **kwargs):
        return self("extract-latent", x, **kwargs)

    def _elbo(self, x, **kwargs):
        raise NotImplementedError

    def _sample(self, num_samples):
        raise NotImplementedError

    def _fixed_sample(self, noise):
        raise NotImplementedError

    def _jvp(self, x, v):
        raise NotImplementedError

    def _ood(self, x):
        raise NotImplementedError

    def _extract_latent(self, x, **kwargs):
        raise NotImplementedError


class BijectionDensity(Density):
    """p(x) = prior(f(x)) |det df/dx|   (exact.py:8-47)."""

    def __init__(self, prior, bijection):
        super().__init__()
        self.bijection = bijection
        self.prior = prior

    def _fused_prehead(self):
        """A run of elementwise bijections ending at the non-square head is one kernel:
        y = a*x + c [-> logit] and one log-jacobian, handed to the head (exact.py:23-30 unrolled)."""
        a, c, logit, node = 1.0, 0.0, False, self
        while isinstance(node, BijectionDensity) and isinstance(node.bijection, _Elementwise):
            if logit:
                return None                       # something follows a logit: not the image pre-processing chain
            bj = node.bijection
            a, c, logit = a * bj.a, c * bj.a + bj.c, bj.logit
            node = node.prior
        return (a, c, logit, node) if isinstance(node, NonSquareHeadDensity) else None

    def _elbo(self, x, **kwargs):
        fused = self._fused_prehead() if x.is_cuda else None
        if fused is not None:
            a, c, logit, head = fused
            E.require_gpu(x)
            y, lj = E.prehead(x.contiguous(), None, a, c, logit)
            return head.elbo(y, _pre_logjac=lj, **kwargs)
        result = self.bijection.x_to_z(x)
        prior_dict = self.prior.elbo(result["z"], **kwargs)
        return {"elbo": prior_dict["elbo"] + result["log-jac"], "bijection-info": result, "prior-dict": prior_dict}

    def _sample(self, num_samples):
        return self.bijection.z_to_x(self.prior.sample(num_samples))["x"]

    def _fixed_sample(self, noise):
        return self.bijection.z_to_x(self.prior.fixed_sample(noise=noise))["x"]

    def _extract_latent(self, x, **kwargs):
        return self.prior.extract_latent(self.bijection.x_to_z(x)["z"], **kwargs)

    def _jvp(self, x, v):
        return self.bijection.jvp(x, v)

    def _ood(self, x):
        return self.prior.ood(self.bijection.x_to_z(x)["z"])


class SplitDensity(Density):
    """Multi-scale split: first half of the channels continues, second half is N(0, I)
    (split.py:6-52).  In non-square models the dropped half is zero-padded on the way back."""

    def __init__(self, density_1, density_2, dim, non_square=False):
        super().__init__()
        self.density_1, self.density_2, self.dim, self.non_square = density_1, density_2, dim, non_square

    def _elbo(self, x):
        x1, x2 = torch.chunk(x, chunks=2, dim=self.dim)
        d1, d2 = self.density_1.elbo(x1.contiguous()), self.density_2.elbo(x2.contiguous())
        return {"elbo": d1["elbo"] + d2["elbo"], "prior-dict": d1, "prior-dict-2": d2}

    def pad_inputs(self, x1):
        return {"x": torch.cat((x1, torch.zeros_like(x1)), dim=self.dim)}

    def _jvp(self, x, v):
        return {"x": self.pad_inputs(x)["x"], "jvp": self.pad_inputs(v)["x"]}

    def _fixed_sample(self, noise):
        x1 = self.density_1.fixed_sample(noise=noise)
        if self.non_square:
            return self.pad_inputs(x1)["x"]
        return torch.cat((x1, self.density_2.fixed_sample(noise=noise)), dim=self.dim)

    def _sample(self, num_samples):
        x1 = self.density_1.sample(num_samples)
        if self.non_square:
            return self.pad_inputs(x1)["x"]
        return torch.cat((x1, self.density_2.sample(num_samples)), dim=self.dim)


class DiagonalGaussianDensity(Density):
    """Standard normal base density (gaussian.py:44-97 with the mean 0 / stddev 1 buffers of
    factory.py:196-201).  Non-unit buffers are honoured by whitening on the host side."""

    def __init__(self, mean, stddev, num_fixed_samples=0):
        super().__init__()
        assert mean.shape == stddev.shape
        self.register_buffer("mean", mean)
        self.register_buffer("stddev", stddev)
        if num_fixed_samples > 0:
            self.register_buffer("_fixed_samples", self._draw(num_fixed_samples))

    @property
    def shape(self):
        return self.mean.shape

    def _draw(self, n):
        eps = torch.randn(n, *self.shape, device=self.mean.device)
        return self.stddev * eps + self.mean

    def logprob_accumulate(self, z, lp):
        """lp[b] += log N(z_b; mean, stddev^2) via the HIP reduction."""
        w = (z - self.mean) / self.stddev if self._nonstandard() else z
        E.gaussian_logprob(w.contiguous(), lp)
        if self._nonstandard():
            lp -= torch.log(self.stddev).sum()

    def _nonstandard(self):
        key = (self.mean._version, self.stddev._version, self.mean.data_ptr())
        if getattr(self, "_ns_cache", (None, None))[0] != key:      # one device read per (re)load, not per call
            self._ns_cache = (key, bool((self.mean != 0).any() or (self.stddev != 1).any()))
        return self._ns_cache[1]

    def _elbo(self, z):
        E.require_gpu(z)
        lp = torch.zeros(z.shape[0], dtype=torch.float32, device=z.device)
        self.logprob_accumulate(z, lp)
        return {"elbo": lp.view(-1, 1), "z": z}

    def _sample(self, num_samples):
        return self._draw(num_samples)

    def _fixed_sample(self, noise):
        return noise if noise is not None else self._fixed_samples

    def _extract_latent(self, x, **kwargs):
        return x


class WrapperDensity(Density):
    def __init__(self, density):
        super().__init__()
        self.density = density

    def _elbo(self, x, **kwargs):
        return self.density.elbo(x, **kwargs)

    def _sample(self, num_samples):
        return self.density.sample(num_samples)

    def _fixed_sample(self, noise):
        return self.density.fixed_sample(noise=noise)

    def _ood(self, x):
        return self.density.ood(x)

    def _extract_latent(self, x, **kwargs):
        return self.density.extract_latent(x, **kwargs)


class DequantizationDensity(WrapperDensity):
    """Adds U[0,1) noise IN PLACE to the caller's tensor, exactly like wrapper.py:28-30."""

    def _elbo(self, x, **kwargs):
        return super()._elbo(x.add_(torch.rand_like(x)), **kwargs)


class DataParallelDensity(nn.Module):
    """Keeps the ``module.`` state-dict prefix of the reference's ``nn.DataParallel`` wrapper
    (wrapper.py:52-68, factory.py:76-81).  Scaling itself is one process per GPU
    (``cmf_amd.distributed``), so this wrapper only forwards."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def elbo(self, x, **kwargs):
        return self("elbo", x, **kwargs)

    def ood(self, x):
        return self("ood", x)

    def extract_latent(self, x, **kwargs):
        return self("extract-latent", x, **kwargs)

    def sample(self, num_samples):
        return self.module.sample(num_samples)

    def fixed_sample(self, noise=None):
        return self.module.fixed_sample(noise=noise)


# --------------------------------------------------------------------------------------------------
# the compiled chain between head and base
# --------------------------------------------------------------------------------------------------


def _scoped(fn):
    """Run a FlowProgram method under its head's KernelConfig (``head.kernels``): which arithmetic the convolution kernels use is
    an attribute of the head, carried by a thread-local scope -- not process state (one thread per GPU, re-entrant)."""
    import functools

    @functools.wraps(fn)
    def run(self, *args, **kwargs):
        with E.scope(self.head.kernels), E._lib.phase(fn.__name__):      # (``head``: a weak proxy, see FlowProgram.__init__)
            return fn(self, *args, **kwargs)
    return run


class FlowProgram:
    """Static plan of the layers between a NonSquareHeadDensity and the Gaussian at the bottom of its
    low-dimensional prior; replaces ``_traverse_backward`` / ``_set_flow_and_jvp_stacks``
    (non_square.py:146-188), which the reference rebuilds on every call."""

    #: bytes of tangent activations one sub-batch may occupy (three hidden panels + J + net output)
    TANGENT_BUDGET = 160 << 30                      # of the 288 GB; C3 at B = 512 needs ~20 GB, C5 at B = 256 ~45 GB

    def __init__(self, head):
        import weakref
        # the head owns the program (``head._program``): holding the head strongly here would make a reference cycle, and heads,
        # parameters and cached buffers would then be freed by the cyclic GC only, not by refcount
        self.head = weakref.proxy(head)
        self.layers = []
        node = head.prior
        while not isinstance(node, NonSquareTailDensity):
            if isinstance(node, BijectionDensity):
                self.layers.append(node.bijection)
                node = node.prior
            elif isinstance(node, SplitDensity):
                self.layers.append(node)
                node = node.density_1
            else:
                raise ValueError(f"Cannot handle density of class {type(node).__name__}")   # non_square.py:168
        self.tail = node
        self.prior = []
        node = node.prior
        while isinstance(node, BijectionDensity):
            self.prior.append(node.bijection)
            node = node.prior
        if not isinstance(node, DiagonalGaussianDensity):
            raise ValueError(f"low-dimensional prior must end in a DiagonalGaussianDensity, got {type(node).__name__}")
        self.gaussian = node
        self.d = self.tail.latent_dimension
        self.image = len(self.tail.x_shape) == 3
        self.layout = "panel" if self.image else "fmajor"
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection) and m.geom.image != self.image:
                raise NotImplementedError("mixed image / flat coupling stacks between head and base are not built")
        for m in self.prior:
            if not isinstance(m, (AffineCouplingBijection, AffineBijection, _ReshapingBijection, _PriorFlowLayer)):
                raise NotImplementedError(f"prior layer {type(m).__name__} is outside the hot path")
        self.zero_in = self._structural_zeros()

    #: False: every coupling layer runs its whole network whatever it is fed (rounds 1 - 4; the A/B switch of the bit-identity tests)
    SKIP_STRUCTURAL_ZEROS = True

    def _structural_zeros(self):
        """Which coupling layers of the DECODE sweep read nothing but structural zeros: {layer index in ``self.layers``: True}.
        Walks the layer list in decode order with a per-element "known zero" map -- set by ``SplitDensity.pad_inputs``
        (split.py:26-30,50-52: the dropped half comes back as zeros, primal and every tangent column alike), carried through
        index maps, cleared on the elements a coupling layer rewrites -- and marks a layer whose network input
        (``pass_elements``) lies wholly inside it.  Static: derived from the layer list, never from data (no host sync).
        The image schemas (schemas.py:399-412 reversed) hit it once: 4 checkerboard couplers -> zero-pad -> a split-channel
        coupler with reverse_mask=True, whose pass-through is the SECOND half of the channels (acl.py:148-160,169-189)."""
        out = {}
        zero = np.zeros(int(np.prod(self.tail.x_shape)), dtype=bool)       # the tail's scatter pattern is data (a buffer): unused
        for i in reversed(range(len(self.layers))):
            m = self.layers[i]
            if isinstance(m, AffineCouplingBijection):
                assert zero.size == m.geom.N
                if m.net.kind == "resnet" and bool(zero[m.pass_elements()].all()):
                    out[i] = True
                zero[m._maps._host["zi"]] = False
            elif isinstance(m, SplitDensity):
                zero = np.concatenate((zero, np.ones_like(zero)))
            elif isinstance(m, _ReshapingBijection) and "z2x" in m._maps._host:
                zero = zero[m._maps._host["z2x"]]               # x[r] = z[z2x[r]]
            # (ViewBijection: a contiguous view, the flat map is unchanged)
        return out

    def _zero(self, i):
        return self.SKIP_STRUCTURAL_ZEROS and self.zero_in.get(i, False)

    #: False: the first decoded coupler's network sees all d Jacobian columns (rounds 1 - 4)
    SEED_COLUMNS = True

    def _seed_columns(self, dev):
        """The FIRST coupling layer of the decode sweep sees the one-hot seed tangents of the tail (non_square.py:303-304, :406-410):
        Jacobian column k is a unit vector at element ``permutation[k]``.  Its network reads only its pass-through elements, and its
        tangent map is linear, so column k's network tangent is identically zero there unless that element is a pass-through one:
        the network runs on THOSE columns only (packed into fewer 16-column slices) and its output is expanded back
        (``cmf_expand_columns``) -- the other columns' rows are scaled by e^{-s}, exactly what the full computation gives them.
        About half the columns on average (a checkerboard layer passes half the elements): C3's recipe model 36 of 64 -> 48 slots,
        C5's 71 of 128 -> 80.  Returns {"index": layer index, "col_of", "colmap": device int32 maps, "nc": packed column slots} or
        None.  The plan depends on the tail's permutation BUFFER (a model constant, saved in checkpoints): read once on the host and
        cached until that buffer changes; nothing here depends on the batch."""
        if not (self.SEED_COLUMNS and self.image and self.layers):
            return None
        first = self.layers[-1]
        if not (isinstance(first, AffineCouplingBijection) and first.net.kind == "resnet") or self.zero_in.get(len(self.layers) - 1):
            return None
        perm = self.tail.permutation
        key = (perm._version, perm.data_ptr(), str(dev))
        hit = getattr(self, "_seed_cols", None)
        if hit is None or hit[0] != key:
            pos = perm[: self.d].detach().cpu().numpy()                 # element of latent k (one device-to-host copy per model)
            is_pass = np.zeros(first.geom.N, dtype=bool)
            is_pass[first.pass_elements()] = True
            cols = [k for k in range(self.d) if is_pass[pos[k]]]
            nc, ncc = E.ceil16(self.d), E.ceil16(max(len(cols), 1))
            plan = None
            if ncc < nc:
                col_of = np.full(first.geom.N, -1, dtype=np.int32)
                colmap = np.full(nc, -1, dtype=np.int32)
                for j, k in enumerate(cols):
                    col_of[pos[k]] = j
                    colmap[k] = j
                plan = {"index": len(self.layers) - 1, "nc": ncc, "n": len(cols), "col_of": torch.from_numpy(col_of).to(dev),
                        "colmap": torch.from_numpy(colmap).to(dev)}
            self._seed_cols = hit = (key, plan)
        return hit[1]

    # -- per-sample tangent footprint, for sub-batching ------------------------------------------
    def tangent_bytes_per_sample(self, nc):
        worst = 0
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection):
                net = m.net
                if net.kind == "resnet":
                    hid = net.module[0].out_channels
                    worst = max(worst, (3 * hid + 2 * m.cmod) * m.geom.HW)
                else:
                    worst = max(worst, 3 * max(l.out_features for l in net if isinstance(l, nn.Linear)))
        D = int(np.prod(self.tail.x_shape))
        return 4 * nc * (worst + 4 * D)

    def train_bytes_per_sample(self, nc, recompute=False):
        """Tangent bytes a training step keeps per sample for the backward pass: every layer input of every coupler network
        (ResNet: 2 K + 1 hidden tensors), the modified rows, the raw output tangent, plus the working set of one layer.
        ``recompute``: only the largest coupler's state at a time, plus one copy of the tangent stack per layer."""
        total, worst = 0, 0
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection):
                net = m.net
                if net.kind == "resnet":
                    hid, nblk = net.module[0].out_channels, sum(1 for b in net.module if hasattr(b, "conv1"))
                    one = ((2 * nblk + 1) * hid + 4 * m.cmod + m.geom.C) * m.geom.HW
                else:
                    one = sum(l.out_features for l in net if isinstance(l, nn.Linear)) + 4 * m.cmod + m.geom.C
                worst = max(worst, one)
                total += m.geom.N if recompute else one
        return 4 * nc * (total + (worst if recompute else 0)) + self.tangent_bytes_per_sample(nc)

    # -- x -> (z_low, low_dim_elbo, earliest latent) ----------------------------------------------
    @_scoped
    def encode(self, x):
        B = x.shape[0]
        h = x.detach().clone().contiguous()
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection):
                m.encode_(h)                     # log-jac above the base is discarded: non_square.py:157-158,177
            elif isinstance(m, SplitDensity):
                n = h[0].numel() // 2            # keep channel half 1 (split.py:16-17)
                idx = torch.arange(n, dtype=torch.int32, device=h.device)
                h = E.gather_primal(h, idx, n).view(B, h.shape[1] // 2, *h.shape[2:])
            else:
                h = m.encode(h)
        z_low = E.gather_primal(h, self.tail.gather_index(h.device), self.d)
        u = z_low.clone()
        lj = torch.zeros(B, dtype=torch.float32, device=x.device)
        for m in self.prior:
            if isinstance(m, (AffineCouplingBijection, AffineBijection)):
                m.encode_(u, lj)
            elif isinstance(m, _PriorFlowLayer):             # nsf prior layers: out of place
                u = m.prior_encode(u, lj)
        self.gaussian.logprob_accumulate(u, lj)
        return z_low, lj, u

    @_scoped
    def encode_nested(self, x):
        """``encode`` that also rebuilds the reference's NESTED ``prior-dict`` (non_square.py:126-129 returns the dict of
        ``self.prior.elbo(x)``: one level per module of the chain -- exact.py:23-30 ``{"elbo", "bijection-info": {"z",
        "log-jac"}, "prior-dict"}``, split.py:15-24 ``{"elbo", "prior-dict", "prior-dict-2"}``, non_square.py:381-395
        ``{"elbo", "low-dim-x", "prior-dict"}``, gaussian.py:65-74 ``{"elbo", "z"}``).  Opt-in (``head.nested_prior_dict``):
        every level costs a copy of the current tensor and a log-jacobian reduction the log-density itself never reads
        (the reference discards them, non_square.py:157-158,177).  Returns (z_low, low_dim_elbo, u, nested dict)."""
        B, dev = x.shape[0], x.device
        zeros = lambda: torch.zeros(B, dtype=torch.float32, device=dev)
        h = x.detach().clone().contiguous()
        levels = []                                  # (kind, contribution to the elbo (B,), payload)
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection):
                lj = zeros()
                m.encode_(h, lj)
                levels.append(("bijection", lj, h.clone()))
            elif isinstance(m, SplitDensity):
                n = h[0].numel() // 2
                h2 = E.gather_primal(h, torch.arange(n, 2 * n, dtype=torch.int32, device=dev), n)
                h2 = h2.view(B, h.shape[1] // 2, *h.shape[2:])
                lp2 = zeros()
                m.density_2.logprob_accumulate(h2, lp2)
                h = E.gather_primal(h, torch.arange(n, dtype=torch.int32, device=dev), n).view(B, h.shape[1] // 2, *h.shape[2:])
                levels.append(("split", lp2, h2))
            else:
                h = m.encode(h)
                levels.append(("bijection", zeros(), h.clone()))
        z_low = E.gather_primal(h, self.tail.gather_index(dev), self.d)
        levels.append(("tail", zeros(), z_low))
        u = z_low.clone()
        for m in self.prior:
            lj = zeros()
            if isinstance(m, (AffineCouplingBijection, AffineBijection)):
                m.encode_(u, lj)
            elif isinstance(m, _PriorFlowLayer):
                u = m.prior_encode(u, lj)
            levels.append(("bijection", lj, u.clone()))
        lp = zeros()
        self.gaussian.logprob_accumulate(u, lp)
        node, acc = {"elbo": lp.view(B, 1), "z": u}, lp
        low_elbo = None
        for kind, c, payload in reversed(levels):
            acc = acc + c
            if kind == "bijection":
                node = {"elbo": acc.view(B, 1), "bijection-info": {"z": payload, "log-jac": c.view(B, 1)}, "prior-dict": node}
            elif kind == "split":
                node = {"elbo": acc.view(B, 1), "prior-dict": node, "prior-dict-2": {"elbo": c.view(B, 1), "z": payload}}
            else:
                low_elbo = acc
                node = {"elbo": acc.view(B, 1), "low-dim-x": payload, "prior-dict": node}
        return z_low, low_elbo, u, node

    def _decode_order(self):
        return [(i, self.layers[i]) for i in reversed(range(len(self.layers)))]

    # -- z_low -> (x_hat, J) ----------------------------------------------------------------------
    @_scoped
    def decode(self, z_low, tangents=True, eps=None):
        B, dev = z_low.shape[0], z_low.device
        N = int(np.prod(self.tail.x_shape))
        scatter = self.tail.scatter_index(dev)
        z = E.gather_primal(z_low.contiguous(), scatter, N).view(B, *self.tail.x_shape)
        T, ncols = None, None
        seed = None
        if tangents:
            ncols = self.d if eps is None else eps.shape[2]
            T = E.seed_tangent(B, N, E.ceil16(ncols), self.layout, scatter, self.d, dev, eps=eps)
            seed = self._seed_columns(dev) if eps is None else None      # (probe directions are dense: every column is live)
        for i, m in self._decode_order():
            if isinstance(m, AffineCouplingBijection):
                m.decode_(z, T, ncols=ncols, zero_in=self._zero(i), seed_columns=seed if seed is not None and seed["index"] == i else None)
            elif isinstance(m, SplitDensity):
                n = z[0].numel()                 # zero-pad the dropped half (split.py:50-52)
                idx = torch.cat((torch.arange(n, dtype=torch.int32, device=dev),
                                 torch.full((n,), -1, dtype=torch.int32, device=dev)))
                z = E.gather_primal(z, idx, 2 * n).view(B, 2 * z.shape[1], *z.shape[2:])
                if T is not None:
                    T = E.gather_tangent(T, idx, 2 * n)
            else:
                z, T = m.decode(z, T)
        return z, T

    # -- reverse sweep: cotangents in data space -> J^T w in latent space ---------------------------
    @_scoped
    def vjp(self, z_low, Wd):
        """``Wd``: (B, D, S) cotangent columns over the flattened data space; returns (x_hat, J^T Wd) with J^T Wd of shape
        (B, d, S) -- the vjp of ``flow_forward`` (non_square.py:190-201) for S directions at once.  A primal decode keeps
        each coupling layer's state, then the layers' adjoints run in encode order; transposed convolutions use the fp32
        MFMA kernel (``engine.net_cotangent``)."""
        B, dev = z_low.shape[0], z_low.device
        N = int(np.prod(self.tail.x_shape))
        z = E.gather_primal(z_low.contiguous(), self.tail.scatter_index(dev), N).view(B, *self.tail.x_shape)
        ctx = []
        for i, m in self._decode_order():
            if isinstance(m, AffineCouplingBijection):
                ctx.append(m.decode_ctx_(z, zero_in=self._zero(i)))
            elif isinstance(m, SplitDensity):
                n = z[0].numel()
                idx = torch.cat((torch.arange(n, dtype=torch.int32, device=dev),
                                 torch.full((n,), -1, dtype=torch.int32, device=dev)))
                z = E.gather_primal(z, idx, 2 * n).view(B, 2 * z.shape[1], *z.shape[2:])
                ctx.append(n)
            else:
                z, _ = m.decode(z, None)
                ctx.append(None)
        S = Wd.shape[2]
        Ct = E.Tangent.from_dense(Wd.contiguous().float(), E.ceil16(S), self.layout)
        for m, c in zip(self.layers, reversed(ctx)):
            if isinstance(m, AffineCouplingBijection):
                m.decode_vjp_(Ct, c)
            elif isinstance(m, SplitDensity):                      # adjoint of the zero-padding: keep the first half
                Ct = E.gather_tangent(Ct, torch.arange(c, dtype=torch.int32, device=dev), c)
            else:
                Ct = m.decode_vjp(Ct)
        out = E.gather_tangent(Ct, self.tail.gather_index(dev), self.d)
        return z, out.to_dense(S).contiguous()

    # -- training: decode with saved state, and its backward -------------------------------
    @_scoped
    def decode_train(self, z_low, tangents=True, keep=True, eps=None, nc=None, nc_hint=None):
        """``decode(z_low, tangents)`` keeping every coupling layer's context (``keep=False``: only its inputs, the rest is
        recomputed layer by layer in ``decode_backward``); returns (x_hat, T, ctx).  ``eps`` (B, d, n): the sweep carries the n
        directions J eps instead of the d Jacobian columns (low-rank Hutchinson backward), in ``nc`` >= n column slots."""
        B, dev = z_low.shape[0], z_low.device
        N = int(np.prod(self.tail.x_shape))
        scatter = self.tail.scatter_index(dev)
        z = E.gather_primal(z_low.contiguous(), scatter, N).view(B, *self.tail.x_shape)
        ncols = self.d if eps is None else eps.shape[2]
        nc = E.ceil16(ncols) if nc is None else int(nc)
        assert nc % 16 == 0 and nc >= ncols
        T = E.seed_tangent(B, N, nc, self.layout, scatter, self.d, dev, eps=eps) if tangents else None
        ctx = []
        for i, m in self._decode_order():
            if isinstance(m, AffineCouplingBijection):
                ctx.append(m.decode_train_(z, T, keep, nc_hint=nc_hint, zero_in=self._zero(i)))
            elif isinstance(m, SplitDensity):
                n = z[0].numel()
                idx = torch.cat((torch.arange(n, dtype=torch.int32, device=dev),
                                 torch.full((n,), -1, dtype=torch.int32, device=dev)))
                z = E.gather_primal(z, idx, 2 * n).view(B, 2 * z.shape[1], *z.shape[2:])
                T = E.gather_tangent(T, idx, 2 * n) if T is not None else None
                ctx.append(n)
            else:
                z, T = m.decode(z, T)
                ctx.append(None)
        return z, T, ctx

    @_scoped
    def decode_tangents_from_ctx(self, ctx, eps=None, nc=None, save=False):
        """A tangent sweep over the PRIMAL contexts of ``decode_train(z_low, tangents=False)``: returns (T, ctx') where ctx' is
        the context list ``decode_backward`` consumes when ``save`` (else None).  Layers are walked like ``decode_train``."""
        dev = self.tail.permutation.device
        B = next(c[0].shape[0] for c in ctx if isinstance(c, tuple))
        N = int(np.prod(self.tail.x_shape))
        ncols = self.d if eps is None else eps.shape[2]
        nc = E.ceil16(ncols) if nc is None else int(nc)
        T = E.seed_tangent(B, N, nc, self.layout, self.tail.scatter_index(dev), self.d, dev, eps=eps)
        out = []
        for (i, m), c in zip(self._decode_order(), ctx):
            if isinstance(m, AffineCouplingBijection):
                out.append(m.decode_tangent_from_ctx_(c, T, save, zero_in=self._zero(i)))
            elif isinstance(m, SplitDensity):
                n = c
                idx = torch.cat((torch.arange(n, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev)))
                T = E.gather_tangent(T, idx, 2 * n)
                out.append(n)
            else:
                T = m.decode_tangent(T)
                out.append(None)
        return T, (out if save else None)

    @_scoped
    def decode_backward(self, ctx, Ct, dx, grads):
        """Backward of ``decode_train``: ``Ct`` = cotangent of the Jacobian stack at the head (e.g. ``engine.gram_backward``),
        ``dx`` = cotangent of x_hat.  Accumulates parameter gradients into ``grads`` and returns the cotangent of z_low (B, d)."""
        B, dev = dx.shape[0], dx.device
        dx = dx.detach().clone().contiguous()
        for m, c in zip(self.layers, reversed(ctx)):
            if isinstance(m, AffineCouplingBijection):
                m.decode_backward_(Ct, dx, c, grads)
            elif isinstance(m, SplitDensity):                      # adjoint of the zero-padding: keep the first half
                if Ct is not None:
                    Ct = E.gather_tangent(Ct, torch.arange(c, dtype=torch.int32, device=dev), c)
                dx = dx.reshape(B, -1)[:, :c].reshape(B, dx.shape[1] // 2, *dx.shape[2:]).contiguous()
            else:
                if Ct is not None:
                    Ct = m.decode_vjp(Ct)
                dx = m.encode(dx)                                  # x[r] = z[z2x[r]]  ->  dz = dx[x2z]
        return E.gather_primal(dx.reshape(B, -1), self.tail.gather_index(dev), self.d)

    @_scoped
    def encode_train(self, x):
        """``encode`` keeping every layer's context: returns (z_low, low_dim_elbo, u, ctx, prior_ctx)."""
        B = x.shape[0]
        h = x.detach().clone().contiguous()
        ctx = []
        for m in self.layers:
            if isinstance(m, AffineCouplingBijection):
                ctx.append(m.encode_train_(h))
            elif isinstance(m, SplitDensity):
                n = h[0].numel() // 2
                idx = torch.arange(n, dtype=torch.int32, device=h.device)
                h = E.gather_primal(h, idx, n).view(B, h.shape[1] // 2, *h.shape[2:])
                ctx.append(n)
            else:
                h = m.encode(h)
                ctx.append(None)
        z_low = E.gather_primal(h, self.tail.gather_index(h.device), self.d)
        u = z_low.clone()
        lj = torch.zeros(B, dtype=torch.float32, device=x.device)
        pctx = []
        for m in self.prior:
            if isinstance(m, _PriorFlowLayer):                     # nsf prior layers: out of place
                u, c = m.prior_encode_train(u, lj)
                pctx.append((m, c))
            elif isinstance(m, (AffineCouplingBijection, AffineBijection)):
                pctx.append((m, m.encode_train_(u, lj)))
        if self.gaussian._nonstandard():
            raise NotImplementedError("training gradients with a non-standard base Gaussian are not built")
        self.gaussian.logprob_accumulate(u, lj)
        return z_low, lj, u, ctx, pctx

    @_scoped
    def prior_backward(self, pctx, u, dlow, grads):
        """Backward of the low-dimensional prior chain: ``dlow`` (B,) = cotangent of low_dim_elbo = log N(u) + sum log-jac;
        returns the cotangent of z_low (B, d) and accumulates the prior flows' parameter gradients."""
        du = torch.empty_like(u)
        dlow = dlow.to(torch.float32).contiguous()
        E._lib.check(E._lib.load().cmf_gaussian_backward(E._p(u.contiguous()), E._p(dlow), u.shape[1], u.shape[0], E._p(du),
                                                         E._stream()), "cmf_gaussian_backward")
        for m, c in reversed(pctx):
            if isinstance(m, _PriorFlowLayer):
                du = m.prior_backward(du, c, grads, dlj=dlow)
            else:
                m.encode_backward_(du, c, grads, dlj=dlow)
        return du

    @_scoped
    def encode_backward(self, ctx, dz_low, grads):
        """Backward of the encode chain above the base: cotangent of z_low -> parameter gradients (and the unused cotangent of x)."""
        B, dev = dz_low.shape[0], dz_low.device
        N = int(np.prod(self.tail.x_shape))
        dh = E.gather_primal(dz_low.contiguous(), self.tail.scatter_index(dev), N).view(B, *self.tail.x_shape)
        for m, c in zip(reversed(self.layers), reversed(ctx)):
            if isinstance(m, AffineCouplingBijection):
                m.encode_backward_(dh, c, grads)
            elif isinstance(m, SplitDensity):                      # the dropped half gets no gradient
                idx = torch.cat((torch.arange(c, dtype=torch.int32, device=dev), torch.full((c,), -1, dtype=torch.int32, device=dev)))
                dh = E.gather_primal(dh, idx, 2 * c).view(B, 2 * dh.shape[1], *dh.shape[2:])
            else:
                dh = m.decode(dh, None)[0]                         # z[r] = x[x2z[r]]  ->  dx = dz[z2x]
        return dh

    # -- latent noise -> z_low (sampling) ----------------------------------------------------------
    @_scoped
    def prior_inverse(self, u):
        z = u.detach().clone().contiguous()
        for m in reversed(self.prior):
            if isinstance(m, (AffineCouplingBijection, AffineBijection)):
                m.decode_(z)
            elif isinstance(m, _PriorFlowLayer):
                z = m.prior_decode(z)
        return z


class NonSquareTailDensity(Density):
    """Bottom of the non-square stack: flatten, fixed random permutation, keep the first d
    coordinates (non_square.py:367-421).  The permutation is drawn at construction and saved in
    checkpoints, like the reference's ``randperm`` buffer (:378)."""

    def __init__(self, prior, x_shape, latent_dimension, detach_before_prior):
        super().__init__()
        self.prior = prior
        self.detach_before_prior = detach_before_prior
        self.x_shape = tuple(x_shape)
        self.latent_dimension = latent_dimension
        self.flattened_dims = int(np.prod(x_shape))
        self.register_buffer("mask", torch.arange(self.flattened_dims) < latent_dimension)
        self.register_buffer("permutation", torch.randperm(self.flattened_dims))
        self.register_buffer("inverse_permutation", torch.argsort(self.permutation))

    def gather_index(self, device):
        return self.permutation[: self.latent_dimension].to(device=device, dtype=torch.int32).contiguous()

    def scatter_index(self, device):
        inv = self.inverse_permutation.to(device)
        return torch.where(inv < self.latent_dimension, inv, torch.full_like(inv, -1)).to(torch.int32).contiguous()

    def _elbo(self, x):
        E.require_gpu(x)
        low = E.gather_primal(x.contiguous(), self.gather_index(x.device), self.latent_dimension)
        prior_dict = self.prior.elbo(low.detach() if self.detach_before_prior else low)
        return {"elbo": prior_dict["elbo"], "low-dim-x": low, "prior-dict": prior_dict}

    def low_dim_to_masked(self, low_dim_x):
        E.require_gpu(low_dim_x)
        out = E.gather_primal(low_dim_x.contiguous(), self.scatter_index(low_dim_x.device), self.flattened_dims)
        return {"x": out.view(low_dim_x.shape[0], *self.x_shape)}

    def _jvp(self, x, v):
        return {"x": self.low_dim_to_masked(x)["x"], "jvp": self.low_dim_to_masked(v)["x"]}

    def _fixed_sample(self, noise):
        return self.low_dim_to_masked(self.prior.fixed_sample(noise))["x"]

    def _sample(self, num_samples):
        return self.low_dim_to_masked(self.prior.sample(num_samples))["x"]

    def _extract_latent(self, x, **kwargs):
        return self.prior.extract_latent(x, **kwargs)


def _cat_nested(outs):
    """Concatenate the result dicts of sub-batches along dim 0, recursing into nested dicts."""
    first = outs[0]
    if isinstance(first, dict):
        return {k: _cat_nested([o[k] for o in outs]) for k in first}
    return torch.cat(outs)


class _ElboFunction(torch.autograd.Function):
    """elbo (B, 1) of a NonSquareHeadDensity as one autograd node over the head's parameters: forward =
    ``train_forward`` (saved state instead of an autograd tape), backward = ``train_backward`` on the HIP kernels."""

    @staticmethod
    def forward(ctx, head, x, kw, pre, box, *params):
        elbo, state = head.train_forward(x, pre_logjac=pre, **kw)
        ctx.head, ctx.state, ctx.params = head, state, params
        ctx.timer = E._timer()                             # autograd calls backward on its own thread
        box["prior-dict"] = state["prior_dict"]            # same keys / nesting as the no-grad path (head.nested_prior_dict too)
        return elbo

    @staticmethod
    def backward(ctx, d_elbo):
        if ctx.state is None:
            raise RuntimeError("cmf_amd: backward through this elbo a second time: the saved tangent state (the bulk of the step's "
                               "memory) is released after the first backward; retain_graph / double backward are not supported -- "
                               "call elbo() again")
        with E._use_timer(ctx.timer):
            grads = ctx.head.train_backward(ctx.state, d_elbo)
        ctx.state = None                                   # the saved tangents are the bulk of the step's memory
        return (None, None, None, None, None, *[grads.get(p) for p in ctx.params])


class NonSquareHeadDensity(Density):
    """log p(x) ~ log p_Z(z) - 1/2 log det(J^T J) - lambda ||x_hat - x||^2 - w_M * g-term
    (non_square.py:22-129), J the Jacobian of the decoder z -> x_hat at z = encode(x)."""

    _VALID_LOG_JACOBIAN_METHODS = ["cholesky", "hutch_with_cg"]
    MAX_ATTEMPTS = 6        # the reference declares it (non_square.py:265) but loops forever; we stop here
    _jacobian_free = False   # M-flow baseline: likelihood term without the log-det (non_square.py:341-346)
    #: False: ``prior-dict`` = the flat ``{"elbo": low_dim_elbo, "low-dim-x": z_low}`` the log-density path itself needs (what
    #: ``_traverse_backward`` digs out of the chain, non_square.py:157-177); True: the reference's full nested chain of dicts
    #: (``FlowProgram.encode_nested``), for callers that walk it
    nested_prior_dict = False
    check_cholesky = "sync"  # "sync": read the retry flags after each call (warn / raise like the reference);
    #                          "lazy": never synchronise; attempts are left in ``last_gram.fail`` on the device

    def __init__(self, prior, regularization_param, log_jacobian_method, x_shape, hutchinson_distribution,
                 num_hutchinson_samples=1, max_cg_iterations=None, cg_tolerance=1):
        super().__init__()
        self.prior = prior
        self.regularization_param = regularization_param
        self.x_shape = tuple(x_shape)
        self.hutchinson_distribution = hutchinson_distribution
        self.num_hutchinson_samples = num_hutchinson_samples
        self.max_cg_iterations = max_cg_iterations
        self.cg_tolerance = cg_tolerance
        if log_jacobian_method not in self._VALID_LOG_JACOBIAN_METHODS:
            raise ValueError(f"{log_jacobian_method} not a valid Jacobian calculation method")
        self.log_jacobian_method = log_jacobian_method
        #: arithmetic of this head's convolution kernels (engine.KernelConfig: tangent "bf16x3" | "f32", primal "f16x3" | "f32" |
        #: "bf16x3"); a plain attribute -- not a parameter or buffer, not in the state dict
        self.kernels = E.KernelConfig()
        self._program = None
        self.last_gram = None

    @property
    def program(self):
        if self._program is None:
            self._program = FlowProgram(self)
        return self._program

    # ------------------------------------------------------------------------------------------
    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())

    def _elbo(self, x, add_reconstruction=True, add_diagonal_metric_reg=False, add_offdiagonal_metric_reg=False,
              likelihood_wt=1., metric_wt=1., visualization=False, ood=False, test_metric=False, _pre_logjac=None):
        E.require_gpu(x)
        if self._wants_grad():
            # autograd mode (the reference's loss.backward(), trainer.py:213): the elbo comes back attached to a custom
            # autograd node whose backward runs the reverse passes on the HIP kernels and hands d loss / d theta to autograd
            if ood or visualization:
                raise NotImplementedError("ood / visualization under autograd: call them under torch.no_grad()")
            params = [p for p in self.parameters() if p.requires_grad]
            kw = dict(add_reconstruction=add_reconstruction, add_diagonal_metric_reg=add_diagonal_metric_reg,
                      add_offdiagonal_metric_reg=add_offdiagonal_metric_reg, likelihood_wt=likelihood_wt, metric_wt=metric_wt)
            box = {}
            elbo = _ElboFunction.apply(self, x, kw, _pre_logjac, box, *params)
            return {"elbo": elbo, "prior-dict": box["prior-dict"]}
        if ood:
            assert self.log_jacobian_method == "cholesky"
        if add_reconstruction:
            assert not visualization
        prog = self.program
        B = x.shape[0]
        want_lik = not np.isclose(likelihood_wt, 0.)
        want_jac = want_lik and not (self._jacobian_free and not (visualization or ood))
        if ood:
            assert add_reconstruction and want_lik
        chunk = B
        if want_jac:
            per = prog.tangent_bytes_per_sample(E.ceil16(prog.d))
            chunk = max(1, min(B, prog.TANGENT_BUDGET // max(per, 1)))
            if chunk < B:
                # equal sub-batches, multiples of 16 where possible: the primal pass groups 16 samples per column slot,
                # and a ragged last sub-batch would run the per-16 paths underfilled
                n = -(-B // chunk)
                chunk = -(-B // n)
                if chunk >= 16:
                    chunk = min(chunk + (-chunk) % 16, B)
        outs = [self._elbo_chunk(x[i:i + chunk], want_lik, want_jac, add_reconstruction, add_diagonal_metric_reg,
                                 add_offdiagonal_metric_reg, likelihood_wt, metric_wt, ood,
                                 None if _pre_logjac is None else _pre_logjac[i:i + chunk])
                for i in range(0, B, chunk)]
        if len(outs) == 1:
            return outs[0]
        return _cat_nested(outs)

    def _elbo_chunk(self, x, want_lik, want_jac, add_rec, add_diag, add_off, lw, mw, ood, pre):
        prog, B, dev = self.program, x.shape[0], x.device
        x = x.contiguous()
        if self.nested_prior_dict:
            z_low, low_elbo, _, prior_dict = prog.encode_nested(x)
        else:
            z_low, low_elbo, _ = prog.encode(x)
            prior_dict = {"elbo": low_elbo.view(B, 1), "low-dim-x": z_low}
        logdet = l1 = None
        if want_jac:
            x_hat, T = prog.decode(z_low, tangents=True)
            hutch = self.training and self.log_jacobian_method == "hutch_with_cg"
            # the Hutchinson branch never factorises (non_square.py:203-258): no jitter may touch its Gram matrix
            g = E.gram_cholesky(T, prog.d, 1 if hutch else self.MAX_ATTEMPTS)
            self.last_gram = g
            if hutch:
                # train mode: Hutchinson + CG surrogate (non_square.py:131-138, :203-258) on the explicit Gram matrix
                self._check_hutchinson_metric(add_off)
                eps = self._hutchinson_probes(B, dev)
                val, u, w, iters = E.hutch_cg(g.jtj, eps, self.max_cg_iterations or prog.d, self.cg_tolerance)
                self.last_hutchinson = {"eps": eps, "u": u, "w": w, "iterations": iters, "value": val}
                logdet = val
                if add_diag or add_off:                       # g-term on the product (J^T J) eps, S == d (non_square.py:87-100)
                    l1_off, l1_diag = E.hutch_metric(w)
                    l1 = l1_diag if add_diag else l1_off
            else:
                self._report_attempts(g)
                logdet = g.logdet
                l1 = g.l1_diag if add_diag else (g.l1_off if add_off else None)
        else:
            x_hat, _ = prog.decode(z_low, tangents=False)      # warm-up: decode only (non_square.py:105-109)
        rec = E.recon_sqerr(x_hat, x) if add_rec else None
        if ood:
            lik = E.elbo_combine(low_elbo, logdet, None, None, None, 1.0, 0.0, 0.0, B, dev)
            return {"likelihood": lik, "reconstruction-error": rec.view(B, 1)}
        elbo = E.elbo_combine(low_elbo if want_lik else None, logdet, rec, l1, pre, lw, self.regularization_param, mw,
                              B, dev)
        return {"elbo": elbo, "prior-dict": prior_dict}

    def _check_hutchinson_metric(self, offdiagonal):
        """The reference reshapes the off-diagonal entries of the (B, d, S) product to (B, d (d - 1)) (non_square.py:98): only
        S == d survives that view.  The DIAGONAL variant (:87-92) takes ``torch.diagonal(dim1=-2, dim2=-1)`` of the rectangular
        block, which is valid for any S and sums min(d, S) entries -- e.g. C5 with S = 4 plus ``g_kk_loss`` trains."""
        if offdiagonal and self.num_hutchinson_samples != self.program.d:
            raise ValueError("off-diagonal metric regularisation with hutch_with_cg needs num_hutchinson_samples == "
                             "latent_dimension (the reference fails at non_square.py:98 otherwise)")

    def _hutchinson_probes(self, B, dev):
        """non_square.py:204-213: N(0,1) or Rademacher probes of shape (B, d, S)."""
        shape = (B, self.program.d, self.num_hutchinson_samples)
        if self.hutchinson_distribution == "normal":
            return torch.randn(*shape, device=dev)
        if self.hutchinson_distribution == "rademacher":
            return torch.bernoulli(0.5 * torch.ones(*shape, device=dev)).mul_(2.).subtract_(1.)
        raise ValueError(f"Unknown hutchinson distribution {self.hutchinson_distribution}")

    def _report_attempts(self, g):
        if self.check_cholesky != "sync":
            return
        fail = g.fail.tolist()                                   # one small D2H copy (synchronises the stream)
        attempts = 1
        while attempts <= self.MAX_ATTEMPTS and fail[attempts - 1]:
            attempts += 1
        if attempts > self.MAX_ATTEMPTS:
            raise RuntimeError(f"J^T J is not positive definite after {self.MAX_ATTEMPTS} jitter attempts "
                               "(the reference would retry forever, non_square.py:280-288)")
        if attempts > 1:
            print(f"WARNING: Numerical non-invertibility in JtJ observed - {attempts} attempts needed to fix")
        g.attempts = attempts

    # ------------------------------------------------------------------------------------------
    def _ood(self, x):
        return self("elbo", x, ood=True)

    def _extract_latent(self, x, **kwargs):
        E.require_gpu(x)
        z_low, _, earliest = self.program.encode(x)
        return earliest if kwargs["earliest_latent"] else z_low

    def _sample(self, num_samples):
        return self._fixed_sample(self.program.gaussian.sample(num_samples))

    def _fixed_sample(self, noise):
        prog = self.program
        u = prog.gaussian.fixed_sample(noise)
        E.require_gpu(u, "latent noise")
        x_hat, _ = prog.decode(prog.prior_inverse(u), tangents=False)
        return x_hat

    # reference-named helpers, kept for callers that reach into the head -----------------------
    def flow_forward(self, z_low):
        return self.program.decode(z_low, tangents=False)[0]

    def jvp_forward(self, z_low, v):
        """One tangent direction v (B, d): returns (x_hat, J v) like non_square.py:322-329."""
        x_hat, T = self.program.decode(z_low, tangents=True, eps=v.reshape(*v.shape, 1).contiguous())
        return x_hat, T.to_dense(1)[:, :, 0].reshape(x_hat.shape)

    def vjp_forward(self, z_low, w):
        """One cotangent direction w (B, *x_shape): returns (x_hat, J^T w) with J^T w of shape (B, d) -- the reverse-mode
        counterpart of ``jvp_forward`` (the reference gets it from autograd, non_square.py:190-201)."""
        x_hat, out = self.program.vjp(z_low, w.reshape(w.shape[0], -1, 1))
        return x_hat, out[:, :, 0]

    def jtj_matvec(self, z_low, v):
        """Matrix-free (J^T J) v for v (B, d, S): one JVP sweep and one VJP sweep (non_square.py:190-201)."""
        x_hat, T = self.program.decode(z_low, tangents=True, eps=v.contiguous())
        S = v.shape[2]
        _, out = self.program.vjp(z_low, T.to_dense(S))
        return x_hat, out

    def jacobian(self, z_low):
        """Dense J (B, D, d) -- the tensor the reference stacks at non_square.py:307."""
        x_hat, T = self.program.decode(z_low, tangents=True)
        return x_hat, T.to_dense(self.program.d).contiguous()


    # ------------------------------------------------------------------------------------------
    # training: forward with saved state, backward on the HIP kernels
    # ------------------------------------------------------------------------------------------
    #: None = automatic: the train-mode Hutchinson backward runs through the 2 S (+ min(d, S)) directions {u_s, eps_s (, e_k)}
    #: instead of all d Jacobian columns whenever that is fewer column slots; False = always differentiate the d-column sweep
    hutch_lowrank = None
    #: column slots of the low-rank sweep are padded to a multiple of this (32: the split-precision weight-gradient kernel
    #: contracts two 16-column slices per MFMA; 16 runs the fp32 weight-gradient kernel on half the columns)
    HUTCH_LOWRANK_NC = 32

    def _hutch_lowrank_columns(self, add_diag=False, add_off=False):
        """(n, nc) of the low-rank sweep -- n directions in nc column slots -- or None where the d-column backward is used:
        the off-diagonal metric term needs S == d (every column anyway), or nothing would be saved."""
        if self.hutch_lowrank is False or add_off:
            return None
        d, S = self.program.d, self.num_hutchinson_samples
        n = 2 * S + (min(d, S) if add_diag else 0)
        q = int(self.HUTCH_LOWRANK_NC)
        nc = -(-n // q) * q
        return (n, nc) if nc < E.ceil16(d) else None

    def head_terms_forward(self, z_low, tangents=True, hutch_eps=None, keep=True, add_diag=False, add_off=False):
        """Decode (with the Jacobian stack) keeping every layer's context, Gram + Cholesky; returns the state
        ``head_terms_backward`` consumes.  ``hutch_eps`` (B, d, S): also run the Hutchinson + CG surrogate of
        non_square.py:203-258 on the explicit Gram matrix (train mode of ``log_jacobian_method = "hutch_with_cg"``).

        Low-rank form (S << d, ``_hutch_lowrank_columns``): the reference builds its graph ONLY through ``w = J^T J eps`` for the
        S probes, ``u`` detached (non_square.py:241-256), so the parameter gradients need the tangents in the 2 S directions
        {u_s, eps_s} alone: dJ = J (M + M^T) = sum_s (J u_s) eps_s^T + (J eps_s) u_s^T with M = 1/S sum_s u_s eps_s^T.  The
        d-column sweep that feeds the explicit Gram matrix then runs WITHOUT saved state, and a second, n-column sweep seeded
        with V = [u | eps (| e_k for the diagonal metric term)] is the one kept for the backward pass."""
        E.require_gpu(z_low)
        with torch.no_grad():
            low = self._hutch_lowrank_columns(add_diag, add_off) if (tangents and hutch_eps is not None) else None
            if low is not None:
                prog, d, S = self.program, self.program.d, hutch_eps.shape[2]
                # ONE primal decode keeping every layer's state; all d columns over it with nothing kept (the Gram matrix) ...
                n, nc = low
                hint = 32 if (nc % 32 == 0 and E.ceil16(d) % 32 == 0) else 16       # what both sweeps' kernels can read
                x_hat, _, pctx = prog.decode_train(z_low.detach(), tangents=False, nc_hint=hint)
                T, _ = prog.decode_tangents_from_ctx(pctx)
                gr = E.gram_cholesky(T, d, 1)                                       # the Hutchinson branch never factorises
                del T
                self.last_gram = gr
                val, u, w, iters = E.hutch_cg(gr.jtj, hutch_eps, self.max_cg_iterations or d, self.cg_tolerance)
                hutch = {"eps": hutch_eps, "u": u, "w": w, "iterations": iters, "value": val, "lowrank": low, "diag": add_diag}
                if add_diag:
                    hutch["l1_off"], hutch["l1_diag"] = E.hutch_metric(w)
                self.last_hutchinson = hutch
                V = torch.zeros(z_low.shape[0], d, n, dtype=torch.float32, device=z_low.device)
                V[:, :, :S] = u
                V[:, :, S:2 * S] = hutch_eps
                if add_diag:
                    k = torch.arange(min(d, S), device=z_low.device)
                    V[:, k, 2 * S + k] = 1.0
                # ... then the n directions over the SAME primal state, this sweep kept for the backward pass
                if keep:
                    TV, ctx = prog.decode_tangents_from_ctx(pctx, eps=V, nc=nc, save=True)
                else:                                                               # recomputation per layer: inputs only
                    del pctx
                    x_hat, TV, ctx = prog.decode_train(z_low.detach(), True, keep, eps=V, nc=nc)
                return {"x_hat": x_hat, "T": TV, "ctx": ctx, "gram": gr, "hutch": hutch}
            x_hat, T, ctx = self.program.decode_train(z_low.detach(), tangents, keep)
            gr = hutch = None
            if tangents:
                # Non-PD batches train on like the reference (non_square.py:280-288, called with create_graph=self.training):
                # the retries enqueued by gram_cholesky add eps I to every sample's G IN PLACE, the jitter is a constant, so
                # the gradient simply flows through the jittered matrix -- which is the ``jtj`` gram_backward inverts.
                gr = E.gram_cholesky(T, self.program.d, 1 if hutch_eps is not None else self.MAX_ATTEMPTS)
                self.last_gram = gr
                if hutch_eps is None:
                    self._report_attempts(gr)               # warn like the reference; raise only after MAX_ATTEMPTS
                if hutch_eps is not None:
                    val, u, w, iters = E.hutch_cg(gr.jtj, hutch_eps, self.max_cg_iterations or self.program.d, self.cg_tolerance)
                    hutch = {"eps": hutch_eps, "u": u, "w": w, "iterations": iters, "value": val}
                    if add_diag or add_off or hutch_eps.shape[2] == self.program.d:
                        hutch["l1_off"], hutch["l1_diag"] = E.hutch_metric(w)
                    self.last_hutchinson = hutch
        return {"x_hat": x_hat, "T": T, "ctx": ctx, "gram": gr, "hutch": hutch}

    def head_terms_backward(self, z_low, x, g_logdet=None, g_l1off=None, g_l1diag=None, g_rec=None, grads=None, state=None):
        """Parameter gradients and the latent cotangent of
            sum_b  g_logdet[b] logdet(J^T J)_b + g_l1off[b] sum_{i!=j}|G_ij| + g_l1diag[b] sum_i |G_ii| + g_rec[b] ||x_hat_b - x_b||^2
        at FIXED z_low -- the head terms of non_square.py:64-129 that depend on the decode side.  ResNet couplers under a
        tangent stack.  Returns a dict with the forward values, ``dz_low`` (B, d) and ``grads`` (parameter -> gradient,
        accumulated into when passed in)."""
        grads = {} if grads is None else grads
        tangents = any(t is not None for t in (g_logdet, g_l1off, g_l1diag))
        st = state if state is not None else self.head_terms_forward(z_low, tangents)
        x_hat, T, gr = st["x_hat"], st["T"], st["gram"]
        B = x_hat.shape[0]
        with torch.no_grad():
            Ct = None
            if tangents and st.get("hutch") is not None:
                # surrogate value_b = mean_s u_s^T (G eps_s) with u detached (the CG solve runs under no_grad in the reference,
                # non_square.py:236-247): d value / d G = mean_s u_s eps_s^T, handed over as an explicit matrix
                # and the metric term on the product W = G eps (S == d): d |W_is| / d G = sign(W_is) e_i eps_s^T
                h = st["hutch"]
                if h.get("lowrank") is not None:
                    # T is the n-column sweep P = J [u | eps (| e_k)]: the objective is a sum of inner products of its columns
                    assert g_l1off is None, "the off-diagonal metric term differentiates all d columns"
                    Cm = E.hutch_lowrank_cotangent(h["w"], h["eps"].shape[2], g_logdet, g_l1diag if h["diag"] else None)
                    Ct = E.gram_backward_matrix(T, Cm)
                else:
                    M = E.hutch_cotangent(h["u"], h["eps"], h["w"], g_logdet, g_l1off, g_l1diag)
                    Ct = E.gram_backward_matrix(T, M)
            elif tangents:
                Ct = E.gram_backward(T, gr.jtj, g_logdet, g_l1off, g_l1diag)
            dx = torch.zeros_like(x_hat)
            if g_rec is not None:
                dx = 2.0 * g_rec.to(torch.float32).view(B, *([1] * (x_hat.dim() - 1))) * (x_hat - x)
            dz = self.program.decode_backward(st["ctx"], Ct, dx, grads)
        out = {"x_hat": x_hat, "dz_low": dz, "grads": grads}
        if gr is not None:
            out.update(logdet=gr.logdet, l1_off=gr.l1_off, l1_diag=gr.l1_diag)
        if st.get("hutch") is not None:
            out["logdet"] = st["hutch"]["value"]
        return out

    #: None = decide from the free HBM; True = always rebuild each coupling layer's tangent state in the backward pass (one more
    #: forward tangent sweep, 1/10 of the memory for a ten-coupler model); False = keep everything
    recompute = None

    def train_forward(self, x, add_reconstruction=True, add_diagonal_metric_reg=False, add_offdiagonal_metric_reg=False,
                      likelihood_wt=1., metric_wt=1., pre_logjac=None):
        """Forward half of a training step: ``elbo`` (B, 1) exactly as ``_elbo`` computes it on the exact path, plus the state
        ``train_backward`` needs (every layer's input, activations, input tangents: ~17 hidden tangent tensors per ResNet
        coupler stay alive -- 131 GB for MNIST d = 64 at the reference's 64 samples per GPU).  ``x`` = the head's input."""
        E.require_gpu(x)
        assert not (add_diagonal_metric_reg and add_offdiagonal_metric_reg)
        prog, B, dev = self.program, x.shape[0], x.device
        want_lik = not np.isclose(likelihood_wt, 0.)
        # M-flow baseline (non_square.py:341-346): the likelihood term is the low-dimensional elbo alone -- no Jacobian, no
        # log-det, and the metric terms have no J^T J to act on (the reference fails on its None)
        want_jac = want_lik and not self._jacobian_free
        if self._jacobian_free and want_lik and (add_diagonal_metric_reg or add_offdiagonal_metric_reg):
            raise ValueError("the M-flow head has no J^T J: metric regularisation is not defined for it (non_square.py:87-100)")
        hutch = want_jac and self.training and self.log_jacobian_method == "hutch_with_cg"     # non_square.py:131-138
        if hutch:
            self._check_hutchinson_metric(add_offdiagonal_metric_reg)
        lowrank = hutch and self._hutch_lowrank_columns(add_diagonal_metric_reg, add_offdiagonal_metric_reg) is not None
        keep = True
        if want_jac:
            # the low-rank Hutchinson backward keeps an n-column sweep (the d-column one runs without saved state first)
            nc = self._hutch_lowrank_columns(add_diagonal_metric_reg, add_offdiagonal_metric_reg)[1] if lowrank else E.ceil16(prog.d)
            free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            keep = (B * prog.train_bytes_per_sample(nc) <= 0.8 * free) if self.recompute is None else not self.recompute
            need = B * max(prog.train_bytes_per_sample(nc, recompute=not keep),
                           prog.tangent_bytes_per_sample(E.ceil16(prog.d)) if lowrank else 0)
            if need > free:
                raise RuntimeError(f"cmf_amd: a training step on {B} samples keeps ~{need / 2**30:.0f} GiB of tangents for the backward "
                                   f"pass (recomputation per coupling layer {'on' if not keep else 'off'}) but {free / 2**30:.0f} GiB "
                                   "are free; use a smaller per-GPU batch (the reference trains with 64 samples per GPU)")
        with torch.no_grad():
            x = x.contiguous()
            z_low, low_elbo, u, ctx, pctx = prog.encode_train(x)
            head = self.head_terms_forward(z_low, tangents=want_jac, hutch_eps=self._hutchinson_probes(B, dev) if hutch else None,
                                           keep=keep, add_diag=add_diagonal_metric_reg, add_off=add_offdiagonal_metric_reg)
            gr = head["gram"]
            rec = E.recon_sqerr(head["x_hat"], x) if add_reconstruction else None
            l1 = logdet = None
            if want_jac:
                src = head["hutch"] if hutch else {"l1_off": gr.l1_off, "l1_diag": gr.l1_diag}
                l1 = src["l1_diag"] if add_diagonal_metric_reg else (src["l1_off"] if add_offdiagonal_metric_reg else None)
                logdet = head["hutch"]["value"] if hutch else gr.logdet
            elbo = E.elbo_combine(low_elbo if want_lik else None, logdet, rec, l1, pre_logjac,
                                  likelihood_wt, self.regularization_param, metric_wt, B, dev)
        if self.nested_prior_dict:
            # opt-in: the reference's nested chain, from one more (no-grad) encode pass like the evaluation path's
            with torch.no_grad():
                prior_dict = prog.encode_nested(x)[3]
        else:
            prior_dict = {"elbo": low_elbo.view(B, 1), "low-dim-x": z_low}
        state = dict(x=x, z_low=z_low, u=u, ctx=ctx, pctx=pctx, head=head, want_lik=want_lik, want_jac=want_jac, rec=add_reconstruction,
                     prior_dict=prior_dict,
                     diag=add_diagonal_metric_reg, off=add_offdiagonal_metric_reg, wl=float(likelihood_wt), wm=float(metric_wt))
        return elbo, state

    def train_backward(self, state, d_elbo, grads=None):
        """Backward half: ``d_elbo`` (B,) or (B, 1) = d loss / d elbo_b (``-1/B`` for ``loss = -elbo.mean()``).  Accumulates
        d loss / d theta of every parameter below this head into ``grads`` (dict parameter -> tensor) and returns it."""
        if grads is None:                                  # one zero-filled pool instead of a fill launch per parameter tensor
            grads = E.GradPool([p for p in self.parameters() if p.requires_grad], d_elbo.device)
        prog = self.program
        w = d_elbo.detach().reshape(-1).to(torch.float32)
        lam = float(self.regularization_param)
        with torch.no_grad():
            # elbo_b = wl (low_b - logdet_b / 2) - lam rec_b - wm l1_b
            lik, jac = state["want_lik"], state["want_jac"]
            out = self.head_terms_backward(state["z_low"], state["x"],
                                           g_logdet=w * (-0.5 * state["wl"]) if jac else None,
                                           g_l1off=w * (-state["wm"]) if jac and state["off"] else None,
                                           g_l1diag=w * (-state["wm"]) if jac and state["diag"] else None,
                                           g_rec=w * (-lam) if state["rec"] else None, grads=grads, state=state["head"])
            dz_low = out["dz_low"]
            if lik:
                dprior = prog.prior_backward(state["pctx"], state["u"], w * state["wl"], grads)
                if not prog.tail.detach_before_prior:
                    dz_low = dz_low + dprior
            prog.encode_backward(state["ctx"], dz_low, grads)
        return grads

    def loss_and_gradients(self, x, pre_logjac=None, grads=None, **kwargs):
        """``loss = -elbo(x, **kw)["elbo"].mean()`` and d loss / d theta for every parameter below this head -- what the reference's
        trainer gets from ``loss.backward()`` (non_square_helpers.py:31-135, trainer.py:207-215).  Returns (loss, elbo, grads)."""
        elbo, state = self.train_forward(x, pre_logjac=pre_logjac, **kwargs)
        B = elbo.shape[0]
        grads = self.train_backward(state, torch.full((B,), -1.0 / B, device=elbo.device), grads)
        return -elbo.mean(), elbo, grads


class ManifoldFlowHeadDensity(NonSquareHeadDensity):
    """Two-step M-flow baseline (non_square.py:341-364): no Jacobian term unless visualising; out of the
    benchmarked path, kept so ``get_non_square_parameters`` finds it by type name."""

    _jacobian_free = True

    def separate_parameters(self, recurse=True):
        likelihood = set(self.program.tail.parameters())
        recon = [p for p in self.parameters() if p not in likelihood]
        return [(p for p in recon), (p for p in likelihood)]
