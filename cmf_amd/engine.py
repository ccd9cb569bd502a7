"""Host-side driver of the HIP kernels: tensors in, kernel launches out.

PyTorch provides device memory (caching allocator) and the current HIP stream; every number is
produced by libcmf_amd.so through the C ABI in include/cmf_amd.h.  There is no CPU path: tensors
that are not on a GPU raise.

Layouts
  primal   (B, N) row-major fp32                       -- identical to the reference's tensors
  tangent  ``Tangent``: T(b, r, col), col = Jacobian column, NC = ceil16(#columns)
             'panel'  (B, N, NC)   image nets: one J panel per sample
             'fmajor' (N, B, NC)   MLP nets: feature-major so that a Linear layer is a 1x1
                                   convolution whose "pixels" are the batch samples
"""
import ctypes as C
import threading

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import (F_NONE, F_RELU, F_TANH, F_RAW, F_SELF_RELU, F_RELU_BITS, O_NONE, O_TANH, O_STANH, ConvPrimalArgs,
                   ConvTangentArgs)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_gpu(t, what="input"):
    if not t.is_cuda:
        raise RuntimeError(f"cmf_amd: {what} is on {t.device}; the log-density path runs only through the HIP "
                           "kernels on an AMD GPU (there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"cmf_amd: {what} must be float32, got {t.dtype}")


def ceil16(n):
    return (int(n) + 15) // 16 * 16


class KernelTimer:
    """Optional live timing of one kernel family with HIP events on the launch stream (bench.py's
    roofline leg).  ``with_events`` brackets a launch with two events; ``summary()`` synchronises once."""

    def __init__(self, select):
        self.select, self.records = select, []

    def wrap(self, name, flops, nbytes, launch):
        if not self.select(name):
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.records.append((name, flops, nbytes, e0, e1))

    def reset(self):
        self.records = []

    def by_name(self):
        """{name: (launches, total_ms, flops, bytes)} over the recorded launches."""
        torch.cuda.synchronize()
        out = {}
        for name, fl, by, e0, e1 in self.records:
            n, ms, f, b = out.get(name, (0, 0.0, 0.0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), f + fl, b + by)
        return out

    def summary(self):
        torch.cuda.synchronize()
        n = len(self.records)
        ms = sum(r[3].elapsed_time(r[4]) for r in self.records)
        return {"launches": n, "total_ms": ms, "flops": sum(r[1] for r in self.records),
                "bytes": sum(r[2] for r in self.records)}


class KernelConfig:
    """Which arithmetic the convolution kernels of a head run in (an attribute of ``NonSquareHeadDensity``: ``head.kernels``;
    every ``FlowProgram`` method runs inside ``scope(head.kernels)``, so two heads -- or two threads -- may differ).

    tangent  3x3 hidden TANGENT convolutions (all d Jacobian columns), their transposes and weight gradients:
             "bf16x3" = split-precision bf16 MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulate; fp32-grade on the tangents: errors
             average over K = 576), "f32" = exact fp32 MFMA.  Layers the split kernel does not cover (1x1, cin % 32 != 0, widths
             other than multiples of 14 / 8) always use the fp32 kernel.
    primal   hidden 3x3 PRIMAL convolutions of a ResNet coupler (16 samples in the column slots of the tangent kernels).  The
             relu masks of the Jacobian are taken from these activations, and a pre-activation that lands on the other side of
             zero than in the reference flips a mask, so this arithmetic has to be fp32-grade PER PRODUCT:
             "f16x3" (default) = fp16 split, 11 + 11 significant bits with exact power-of-two operand scales (~2^-22 per product:
             as few mask flips as exact fp32 products, profiles/r04_primal_precision_study.txt) at the bf16 MFMA rate;
             "f32" = exact fp32 MFMA (2.8x slower); "bf16x3" = bf16 split (~2^-16: ~15x more mask flips, median per-sample
             log-det / g_ij error 3e-6 / 1.4e-5 instead of 1e-6 / 2e-7 -- kept as an experiment, never the default)."""
    __slots__ = ("tangent", "primal")

    def __init__(self, tangent="bf16x3", primal="f16x3"):
        assert tangent in ("bf16x3", "f32") and primal in ("f16x3", "f32", "bf16x3")
        self.tangent, self.primal = tangent, primal

    def __repr__(self):
        return f"KernelConfig(tangent={self.tangent!r}, primal={self.primal!r})"


_DEFAULT_CONFIG = KernelConfig()
_tls = threading.local()


def cfg():
    """The calling thread's current KernelConfig (the innermost ``scope``), or the defaults."""
    stack = getattr(_tls, "cfg", None)
    return stack[-1] if stack else _DEFAULT_CONFIG


class scope:
    """``with scope(config):`` -- kernels launched by THIS thread inside the block use ``config`` (re-entrant, nestable)."""

    def __init__(self, config=None, **kw):
        cur = cfg()
        self.config = config if config is not None else KernelConfig(
            **{**{"tangent": cur.tangent, "primal": cur.primal}, **kw})

    def __enter__(self):
        if not hasattr(_tls, "cfg"):
            _tls.cfg = []
        _tls.cfg.append(self.config)
        return self.config

    def __exit__(self, *exc):
        _tls.cfg.pop()
        return False


def _timer():
    return getattr(_tls, "timer", None)


class _use_timer:
    """Install an existing KernelTimer (or None) on the calling thread: autograd runs ``backward`` on its own thread, and the
    timer that was active when the elbo was computed should see the backward kernels too (tools/bench_train.py)."""

    def __init__(self, timer):
        self.timer = timer

    def __enter__(self):
        self.prev = _timer()
        _tls.timer = self.timer
        return self.timer

    def __exit__(self, *exc):
        _tls.timer = self.prev
        return False


class timing:
    """``with timing(select) as t:`` -- launches of THIS thread whose family name ``select`` accepts are bracketed by HIP
    events on the launch stream; ``t.by_name()`` / ``t.summary()`` synchronise once (bench.py's roofline / stages legs)."""

    def __init__(self, select):
        self.timer = KernelTimer(select)

    def __enter__(self):
        self.prev = _timer()
        _tls.timer = self.timer
        return self.timer

    def __exit__(self, *exc):
        _tls.timer = self.prev
        return False


def _use_bf16x3(taps, cin, W, transpose, H=None, cout=64):
    return cfg().tangent == "bf16x3" and _shape_ok_bf16x3(taps, cin, W, transpose, H, cout)


def _shape_ok_bf16x3(taps, cin, W, transpose, H=None, cout=64):
    """Shapes the split-precision kernel is built for: whole 2 x 14 (or 2 x 8) pixel tiles, whole 64- (or one 32-) channel groups."""
    return (taps == 9 and cin % 8 == 0 and cin % 32 == 0 and ((W % 14 == 0 and (H is None or H % 2 == 0)) or (W % 8 == 0 and (H is None or H % 4 == 0)))
            and (cout % 64 == 0 or cout == 32))


class Tangent:
    """A stack of Jacobian columns attached to a primal tensor of N elements per sample."""

    def __init__(self, B, N, nc, layout, device, data=None):
        self.B, self.N, self.nc, self.layout = int(B), int(N), int(nc), layout
        numel = self.B * self.N * self.nc
        self.data = data if data is not None else torch.empty(numel, dtype=torch.float32, device=device)
        assert self.data.numel() >= numel

    @property
    def t_b(self):
        return self.N * self.nc if self.layout == "panel" else self.nc

    @property
    def t_r(self):
        return self.nc if self.layout == "panel" else self.B * self.nc

    def like(self, N):
        return Tangent(self.B, N, self.nc, self.layout, self.data.device)

    @staticmethod
    def from_dense(dense, nc, layout):
        """(B, N, S) torch tensor -> tangent / cotangent stack with S <= nc columns (the rest zero)."""
        B, N, S = dense.shape
        t = Tangent(B, N, nc, layout, dense.device)
        buf = t.data[: B * N * nc].zero_()
        if layout == "panel":
            buf.view(B, N, nc)[:, :, :S] = dense
        else:
            buf.view(N, B, nc)[:, :, :S] = dense.permute(1, 0, 2)
        return t

    def to_dense(self, ncols):
        """(B, N, ncols) torch view/copy for tests and the public jvp API."""
        if self.layout == "panel":
            return self.data[: self.B * self.N * self.nc].view(self.B, self.N, self.nc)[:, :, :ncols]
        return self.data[: self.B * self.N * self.nc].view(self.N, self.B, self.nc).permute(1, 0, 2)[:, :, :ncols]


# --------------------------------------------------------------------------------------------------
# packed weights, cached per parameter version
# --------------------------------------------------------------------------------------------------


class _PackCache:
    """Packed (kernel-layout) copies of weights, rebuilt when the parameter is updated in place
    (``_version``), re-allocated or moved.  Entries hold a weak reference to the parameter and are
    validated by identity: ``id()`` values and device addresses are recycled after garbage collection."""

    def __init__(self):
        self._store = {}
        self.generation = 0
        self._tables = {}                                          # device -> (signature, descriptor table) of the batched refresh
        self._lock = threading.RLock()                             # two threads on one device may share the cache
        self._refreshed = {}                                       # device -> generation its batched refresh last ran for

    def invalidate(self):
        """Every cached pack is stale from now on.  Called by writers that change parameter VALUES without touching the
        tensors' version counters: ``FlatOptimizer.step`` updates the flat buffer with a raw HIP kernel, and a parameter
        whose ``.data`` is a view of that buffer keeps ``_version`` 0 whatever happens to the storage."""
        self.generation += 1

    def get(self, weight, taps, transpose=False, bf16x3=False):
        """``bf16x3``: False = cmf_pack_weight's floats, True / "bf16x3" = the split-precision pack, "f16x3" = the fp16 pack
        (scaled by a power of two, 16-byte trailer).  Thread-safe: the whole lookup / rebuild runs under the cache's lock, and a
        pack built by a launch on ANOTHER stream is ordered before the caller's stream through the event recorded behind that
        launch (two threads on one device with a stream each)."""
        with self._lock:
            out, order = self._get(weight, taps, transpose, bf16x3)
            self._wait(order)
            return out

    @staticmethod
    def _mark():
        """(stream id, event recorded on it now): what a later consumer on a different stream has to wait for."""
        if torch.cuda.is_current_stream_capturing():
            return None
        st = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(st)
        return [st.cuda_stream, ev]

    @staticmethod
    def _wait(order):
        if order is None or order[1] is None:                        # (no launch to order against, or known complete)
            return
        if torch.cuda.is_current_stream_capturing():                  # (capture follows warm-up calls on the capture stream's side)
            return
        if order[1].query():
            order[1] = None                                           # complete for every stream from now on: no more host calls
            return
        st = torch.cuda.current_stream()
        if st.cuda_stream != order[0]:
            st.wait_event(order[1])

    def _get(self, weight, taps, transpose, bf16x3):
        import weakref
        bf16x3 = {False: 0, True: 1, "bf16x3": 1, "f16x3": 2, 0: 0, 1: 1, 2: 2}[bf16x3]      # = cmf_pack_desc.kind
        key = (id(weight), bool(transpose), bf16x3)
        ver = (weight._version, weight.data_ptr(), weight.device, tuple(weight.shape), self.generation)
        hit = self._store.get(key)
        if hit is not None and hit[0]() is weight and hit[1] == ver:
            return hit[2], hit[4]
        if (hit is not None and hit[0]() is weight and hit[1][:4] == ver[:4] and self.BATCHED_REFRESH
                and self._refreshed.get(weight.device) != self.generation and not torch.cuda.is_current_stream_capturing()):
            # only the generation moved: an optimiser step rewrote the parameter VALUES (FlatOptimizer.step -> invalidate()).
            # Every pack of the model is stale in the same way: rebuild them all in ONE launch instead of one launch each --
            # once per generation and device (entries it had to skip are rebuilt singly below), never inside a graph capture
            # (the descriptor table is a host -> device copy)
            self._refresh_generation(weight.device)
            self._refreshed[weight.device] = self.generation
            hit = self._store.get(key)
            if hit[1] == ver:
                return hit[2], hit[4]
        lib = _lib.load()
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        n = C.c_longlong(0)
        w = weight.detach().contiguous()
        # a stale entry's buffer is reused when the size still fits (training: every pack is rebuilt after every optimiser step)
        old = hit[2] if hit is not None and hit[0]() is weight and hit[2].device == weight.device else None
        buf = lambda nbytes, dt: old if old is not None and old.dtype == dt and old.numel() * old.element_size() == nbytes \
            else torch.empty(nbytes // torch.empty(0, dtype=dt).element_size(), dtype=dt, device=weight.device)
        if bf16x3:
            if transpose:                                  # the adjoint operator: channels swapped, taps flipped -- by the pack kernel
                cout, cin = cin, cout
            fn, what = ((lib.cmf_pack_weight_bf16x3_t, "cmf_pack_weight_bf16x3_t") if bf16x3 == 1 else
                        (lib.cmf_pack_weight_f16x3, "cmf_pack_weight_f16x3"))
            _lib.check(fn(None, None, cout, cin, int(transpose), C.byref(n), None), "pack size")
            out = buf(n.value, torch.uint8)
            _lib.check(fn(_p(w), _p(out), cout, cin, int(transpose), None, _stream()), what)
        else:
            _lib.check(lib.cmf_pack_weight(None, None, cout, cin, taps, int(transpose), C.byref(n), None), "pack size")
            out = buf(4 * n.value, torch.float32)
            _lib.check(lib.cmf_pack_weight(_p(w), _p(out), cout, cin, taps, int(transpose), None, _stream()), "cmf_pack_weight")
        if len(self._store) > 4096:                                   # drop entries whose parameter is gone
            self._store = {k: v for k, v in self._store.items() if v[0]() is not None}
        order = self._mark()
        self._store[key] = (weakref.ref(weight), ver, out, int(taps), order)
        return out, order

    #: False: every stale pack is rebuilt by its own launch on first use (rounds 1 - 2)
    BATCHED_REFRESH = True

    def _refresh_generation(self, device):
        """Re-pack, in one ``cmf_pack_weights_batched`` launch, every entry on ``device`` whose parameter is alive and unchanged
        but for the generation counter (its values were rewritten under it by the fused optimiser step)."""
        todo = []
        for key, (ref, ver, out, taps, _) in list(self._store.items()):
            w = ref()
            if (w is None or ver[4] == self.generation or w.device != device or not w.is_contiguous()
                    or (w._version, w.data_ptr(), w.device, tuple(w.shape)) != ver[:4]):
                continue
            if not isinstance(w, nn.Parameter):
                # a DERIVED tensor (masked MADE weight, L U product: engine.DERIVED): its values are still the old ones until its
                # own rebuild copies the new ones in (bumping _version); re-packing it now would stamp a stale pack as fresh
                continue
            todo.append((key, w, ver, out, taps))
        if not todo:
            return
        desc = np.dtype([("w", "<u8"), ("out", "<u8"), ("total", "<i8"), ("cout", "<i4"), ("cin", "<i4"), ("taps", "<i4"),
                         ("transpose", "<i4"), ("kind", "<i4"), ("reserved", "<i4")])            # = cmf_pack_desc (include/cmf_amd.h)
        sig = tuple((key, w.data_ptr(), out.data_ptr()) for key, w, ver, out, taps in todo)
        cached = self._tables.get(device)
        if cached is not None and cached[0] == sig:
            table = cached[1]
        else:
            arr = np.zeros(len(todo), dtype=desc)
            for i, (key, w, ver, out, taps) in enumerate(todo):
                _, transpose, bf16x3 = key
                cout, cin = int(w.shape[0]), int(w.shape[1])
                if bf16x3 and transpose:
                    cout, cin = cin, cout
                total = out.numel() * out.element_size()
                total = (total - 16) // 2 if bf16x3 == 2 else total // (2 if bf16x3 else 4)     # kind 2: 16-byte trailer
                arr[i] = (w.data_ptr(), out.data_ptr(), total, cout, cin, taps, int(transpose), int(bf16x3), 0)
            table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
            self._tables[device] = (sig, table)
        _lib.check(_lib.load().cmf_pack_weights_batched(_p(table), len(todo), _stream()), "cmf_pack_weights_batched")
        order = self._mark()                                          # one event behind the one launch, shared by its entries
        for key, w, ver, out, taps in todo:
            self._store[key] = (self._store[key][0], ver[:4] + (self.generation,), out, taps, order)


PACKS = _PackCache()


# --------------------------------------------------------------------------------------------------
# thin wrappers over the C ABI
# --------------------------------------------------------------------------------------------------


def conv_primal(x_ptr_t, x_off, x_b, x_c, x_px, weight, taps, bias, y, y_b, y_c, y_px, B, cin, cout, H, W,
                imode=F_NONE, mask=None, f_c=0, f_px=0, omode=O_NONE, sw=None, sb=None, g=None, res=None):
    """One primal conv/linear launch.  ``x_ptr_t`` is the tensor holding the input, ``x_off`` an element offset."""
    lib = _lib.load()
    a = ConvPrimalArgs()
    a.x = C.c_void_p(x_ptr_t.data_ptr() + 4 * int(x_off)); a.x_b, a.x_c, a.x_px = int(x_b), int(x_c), int(x_px)
    a.f = _p(mask); a.f_c, a.f_px = int(f_c), int(f_px); a.imode = imode
    a.w = _p(PACKS.get(weight, taps)); a.bias = _p(bias); a.sw = _p(sw); a.sb = _p(sb)
    a.y = _p(y); a.y_b, a.y_c, a.y_px = int(y_b), int(y_c), int(y_px)
    a.g = _p(g)
    a.r = _p(res); a.r_b, a.r_c, a.r_px = int(y_b), int(y_c), int(y_px)
    a.omode = omode
    a.B, a.cin, a.cout, a.H, a.W, a.taps = int(B), int(cin), int(cout), int(H), int(W), int(taps)
    _lib.check(lib.cmf_conv_primal(C.byref(a), _stream()), "cmf_conv_primal")


def conv_tangent(x_t, x_off, x_np, x_ci, x_px, weight, taps, y_t, y_np, y_co, y_px, np_, cin, cout, H, W, nc,
                 fmode=F_NONE, f=None, f_np=0, f_ci=0, f_px=0, res_t=None, transpose=False, bias=None, f_group=1,
                 x_sl=16, y_sl=16, precision=None, y_off=0, res_off=0, fo=None, fo_np=0, fo_co=0, fo_px=0, fomode=F_NONE,
                 mask_out=None, mask_np=0, amax_in=None, amax_out=None, item_channels=0, live=0, res_np=None):
    """``fo`` = OUTPUT-side factor (reverse sweep, fp32 kernel only); ``y_off`` / ``res_off`` = element offsets into
    ``y_t`` / ``res_t`` (in-place accumulation into a strided view of a larger tensor).  ``precision`` "f16x3": the fp16 split
    kernel of the primal pass (``amax_in`` / ``amax_out``: one-float device tensors, the input-range chain).  ``live`` 1 / 2:
    checkerboard output (split-precision kernel only): the pixels with (row + col) % 2 == live - 1, stored compactly
    (``res_np``: sample stride of the residual, a FULL image then; the other residual strides are y's)."""
    lib = _lib.load()
    a = ConvTangentArgs()
    a.x = C.c_void_p(x_t.data_ptr() + 4 * int(x_off)); a.x_np, a.x_ci, a.x_px = int(x_np), int(x_ci), int(x_px)
    a.f = _p(f); a.f_np, a.f_ci, a.f_px = int(f_np), int(f_ci), int(f_px); a.fmode = fmode
    # split-precision kernel: any input factor without output factor, or NO input factor with an optional relu' BIT MASK on the
    # output (no residual then): the transposed convs of the reverse sweep
    obits = isinstance(fo, BitMask)
    # ... or with the residual IN PLACE (res_t is y_t at the same offset: y <- y + mask . conv(x), the reverse sweep's skip connection)
    inplace = res_t is not None and res_t.data_ptr() == y_t.data_ptr() and int(res_off) == int(y_off)
    precision = precision or cfg().tangent
    split = (precision == "bf16x3" and _shape_ok_bf16x3(taps, cin, W, transpose, H, cout)
             and not (bias is not None and cout > 64)
             and ((fmode != F_NONE and fo is None) or
                  (fmode == F_NONE and (fo is None or (obits and (res_t is None or inplace) and cout % 64 == 0)))))
    # fp16 split kernel: the primal pass's forward form (the input's own relu) or its backward form (plain cotangent in, per-column
    # relu' of a float activation laid out like y on the way out, optional residual: net_primal_backward)
    pbwd = fmode == F_NONE and fo is not None and not obits and fomode == F_SELF_RELU
    # (the split kernels fetch their per-channel constants once per launch: a bias only with a single 64-channel group)
    f16 = (precision == "f16x3" and cout % 64 == 0 and _shape_ok_bf16x3(taps, cin, W, transpose, H, cout)
           and not (bias is not None and cout > 64)
           and ((fmode == F_SELF_RELU and fo is None and not transpose) or (pbwd and bias is None and mask_out is None)))
    assert not obits or split, "bit-mask output factors are applied by the split-precision kernel only"
    if obits:
        fo, fo_np, fo_co, fo_px, fomode = fo.data, fo.np_bytes, 0, 0, F_RELU_BITS
    a.w = _p(PACKS.get(weight, taps, transpose, bf16x3="f16x3" if f16 else split))
    a.y = C.c_void_p(y_t.data_ptr() + 4 * int(y_off)); a.y_np, a.y_co, a.y_px = int(y_np), int(y_co), int(y_px)
    a.r = None if res_t is None else C.c_void_p(res_t.data_ptr() + 4 * int(res_off))
    a.r_np, a.r_co, a.r_px = int(y_np if res_np is None else res_np), int(y_co), int(y_px)
    a.fo = _p(fo); a.fo_np, a.fo_co, a.fo_px, a.fomode = int(fo_np), int(fo_co), int(fo_px), int(fomode)
    a.mask_out = _p(mask_out); a.mask_np = int(mask_np)
    assert not (fmode == F_RELU_BITS and not split), "bit-mask factors are read by the split-precision kernel only"
    assert not (mask_out is not None and split), "sign bits are written by the fp32 and the fp16-split kernels only"
    a.amax_in, a.amax_out = (_p(amax_in), _p(amax_out)) if f16 else (None, None)
    a.live = int(live)
    assert not live or split, "checkerboard output is the split-precision kernel's"
    a.np, a.cin, a.cout, a.H, a.W, a.nc, a.taps = int(np_), int(cin), int(cout), int(H), int(W), int(nc), int(taps)
    a.bias = _p(bias); a.f_group = int(f_group)
    a.x_sl, a.y_sl, a.r_sl = int(x_sl), int(y_sl), int(y_sl)
    fn, what = ((lib.cmf_conv_tangent_f16x3, "cmf_conv_tangent_f16x3") if f16 else
                (lib.cmf_conv_tangent_bf16x3, "cmf_conv_tangent_bf16x3") if split else (lib.cmf_conv_tangent, "cmf_conv_tangent"))
    launch = lambda: _lib.check(fn(C.byref(a), _stream()), what)
    if f16 and item_channels:                               # tests / probes: the item size cmf_conv_tangent_f16x3 would pick itself
        launch = lambda: _lib.check(lib.cmf_conv_tangent_f16x3_item(C.byref(a), int(item_channels), _stream()), "cmf_conv_tangent_f16x3_item")
    if fmode == F_SELF_RELU and not f16 and _lib.tracing():
        run = launch

        def launch():                                       # traced: a primal launch of a tangent kernel (bench.py's stages)
            with _lib.role("primal"):
                run()
    TIMER = _timer()
    if TIMER is None:
        return launch()
    # algorithmic work of this launch: 2*cin*cout*taps FLOP per output pixel and Jacobian column; every input,
    # output and residual element crosses HBM once (4 bytes each)
    px_in = float(H) * W * nc * np_
    px = px_in * (0.5 if live else 1.0)                     # checkerboard output: half the output pixels (and residual reads), every input pixel
    TIMER.wrap(f"conv_tangent_t{taps}_ci{cin}_co{cout}" + ("_primal" if fmode == F_SELF_RELU else "_primal_bwd" if pbwd else "_live" if live else ""),
               2.0 * cin * cout * taps * px,
               4.0 * (px_in * cin + px * (cout + (cout if res_t is not None else 0))), launch)


_WGRAD_WS = {}


def conv_tangent_wgrad(x_t, x_off, x_np, x_ci, x_px, gy_t, gy_off, y_np, y_co, y_px, dw, taps, np_, cin, cout, H, W, nc,
                       fmode=F_NONE, f=None, f_np=0, f_ci=0, f_px=0, f_group=1, x_sl=16, y_sl=16, precision=None):
    """dw (cout, cin, kh, kw) += weight gradient of the tangent conv described like ``conv_tangent``'s forward launch;
    ``gy_t`` (+ ``gy_off`` elements) is the cotangent of y with y's strides.  fp32 factor tensors only."""
    lib = _lib.load()
    assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * cin * taps
    a = ConvTangentArgs()
    a.x = C.c_void_p(x_t.data_ptr() + 4 * int(x_off)); a.x_np, a.x_ci, a.x_px = int(x_np), int(x_ci), int(x_px)
    fbits = isinstance(f, BitMask)                                   # relu' as a bit mask: the split kernel only
    if fbits:
        f, f_np, f_ci, f_px, fmode = f.data, f.np_bytes, 0, 0, F_RELU_BITS
    a.f = _p(f); a.f_np, a.f_ci, a.f_px = int(f_np), int(f_ci), int(f_px); a.fmode = fmode
    a.y_np, a.y_co, a.y_px = int(y_np), int(y_co), int(y_px)
    a.np, a.cin, a.cout, a.H, a.W, a.nc, a.taps = int(np_), int(cin), int(cout), int(H), int(W), int(nc), int(taps)
    a.f_group = int(f_group)
    a.x_sl, a.y_sl = int(x_sl), int(y_sl)
    need = int(lib.cmf_conv_tangent_wgrad_ws(C.byref(a)))
    key = (x_t.device, torch.cuda.current_stream().cuda_stream)      # one workspace per device and stream (re-entrant)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _WGRAD_WS[key] = torch.empty(need // 4, dtype=torch.float32, device=x_t.device)
    gy = C.c_void_p(gy_t.data_ptr() + 4 * int(gy_off))
    # split-precision kernel (operands shared through LDS) where the forward convs use one: whole 64-channel blocks, column pairs
    split = ((precision or cfg().tangent) == "bf16x3" and taps == 9 and cin % 64 == 0 and cout % 64 == 0 and nc % 32 == 0
             and fmode in (F_NONE, F_RELU, F_SELF_RELU, F_RELU_BITS) and f_group <= 1)
    assert not fbits or split, "bit-mask factors are read by the split-precision weight-gradient kernel only"
    fn, what = (lib.cmf_conv_tangent_wgrad_bf16x3, "cmf_conv_tangent_wgrad_bf16x3") if split else \
               (lib.cmf_conv_tangent_wgrad, "cmf_conv_tangent_wgrad")
    launch = lambda: _lib.check(fn(C.byref(a), gy, _p(dw), _p(ws), need, _stream()), what)
    TIMER = _timer()
    if TIMER is None:
        return launch()
    px = float(H) * W * nc * np_
    TIMER.wrap(f"conv_wgrad_t{taps}_ci{cin}_co{cout}" + ("_primal" if fmode == F_SELF_RELU else ""), 2.0 * cin * cout * taps * px,
               4.0 * px * (cin + cout), launch)


def conv_tangent_wgrad_batched(xs, gys, dws, x_np, x_ci, x_px, y_np, y_co, y_px, np_, cin, cout, H, W, nc, fmode=F_NONE, x_sl=16, y_sl=16):
    """``conv_tangent_wgrad`` for up to 16 problems of ONE shape (lists of input tensors, cotangent tensors and weight-gradient
    tensors) in one launch of the split-precision kernel: the 256 persistent workgroups are split over the problems.  For launches
    that are a few dozen image rows each (the primal weight gradients of a coupler at a training shard's sample groups)."""
    lib = _lib.load()
    n = len(xs)
    assert n == len(gys) == len(dws) and 1 <= n <= _lib.WGRAD_MAX_BATCH and fmode in (F_NONE, F_SELF_RELU)
    assert cin % 64 == 0 and cout % 64 == 0 and nc % 32 == 0
    a = ConvTangentArgs()
    a.x = _p(xs[0]); a.x_np, a.x_ci, a.x_px = int(x_np), int(x_ci), int(x_px)
    a.fmode = fmode
    a.y_np, a.y_co, a.y_px = int(y_np), int(y_co), int(y_px)
    a.np, a.cin, a.cout, a.H, a.W, a.nc, a.taps = int(np_), int(cin), int(cout), int(H), int(W), int(nc), 9
    a.f_group = 1
    a.x_sl, a.y_sl = int(x_sl), int(y_sl)
    need = int(lib.cmf_conv_tangent_wgrad_ws(C.byref(a)))
    key = (xs[0].device, torch.cuda.current_stream().cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _WGRAD_WS[key] = torch.empty(need // 4, dtype=torch.float32, device=xs[0].device)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    for dw in dws:
        assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == cout * cin * 9
    launch = lambda: _lib.check(lib.cmf_conv_tangent_wgrad_bf16x3_batched(C.byref(a), n, arr(xs), arr(gys), arr(dws), _p(ws), need, _stream()),
                                "cmf_conv_tangent_wgrad_bf16x3_batched")
    TIMER = _timer()
    if TIMER is None:
        return launch()
    px = float(H) * W * nc * np_ * n
    TIMER.wrap(f"conv_wgrad_t9_ci{cin}_co{cout}" + ("_primal" if fmode == F_SELF_RELU else "") + "_batched", 2.0 * cin * cout * 9 * px,
               4.0 * px * (cin + cout), launch)


def primal_regroup(t, to_grouped):
    """(B, N) <-> (B/16, N, 16): sample-grouped layout used to run primal data through the tangent kernels."""
    B = t.shape[0] if to_grouped else t.shape[0] * 16
    N = t[0].numel() if to_grouped else t[0].numel() // 16
    out = torch.empty(t.numel(), dtype=torch.float32, device=t.device)
    _lib.check(_lib.load().cmf_primal_regroup(_p(t), _p(out), B, N, int(to_grouped), _stream()), "cmf_primal_regroup")
    return out


def absmax(t, out):
    """out[0] = max(out[0], max |t|) for a flat fp32 tensor (the input-range word of the fp16-split primal convs)."""
    _lib.check(_lib.load().cmf_absmax(_p(t), t.numel(), _p(out), _stream()), "cmf_absmax")


def gather_primal(src, idx, n_out, out=None):
    """out(b, r) = idx[r] >= 0 ? src(b, idx[r]) : 0.  src: (B, N...) contiguous."""
    B = src.shape[0]
    src2 = src.reshape(B, -1)
    if out is None:
        out = torch.empty(B, n_out, dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().cmf_gather_primal(_p(src2), src2.shape[1], _p(out), n_out, _p(idx), n_out, B, _stream()),
               "cmf_gather_primal")
    return out


def gather_tangent(T, idx, n_out):
    out = T.like(n_out)
    _lib.check(_lib.load().cmf_gather_tangent(_p(T.data), T.t_b, T.t_r, _p(out.data), out.t_b, out.t_r, _p(idx), n_out,
                                              T.nc, T.B, _stream()), "cmf_gather_tangent")
    return out


def expand_columns(T, nc_out, colmap):
    """A panel-layout tangent stack with ``nc_out`` column slots: column c = column ``colmap[c]`` of ``T`` (or zero where -1)."""
    assert T.layout == "panel"
    out = Tangent(T.B, T.N, nc_out, "panel", T.data.device)
    _lib.check(_lib.load().cmf_expand_columns(_p(T.data), T.nc, _p(out.data), int(nc_out), _p(colmap), T.B * T.N, _stream()),
               "cmf_expand_columns")
    return out


def seed_tangent(B, N, nc, layout, col_of, d, device, eps=None):
    T = Tangent(B, N, nc, layout, device)
    S = 0 if eps is None else int(eps.shape[2])
    _lib.check(_lib.load().cmf_seed_tangent(_p(T.data), T.t_b, T.t_r, _p(col_of), N, nc, _p(eps), d, S, B, _stream()),
               "cmf_seed_tangent")
    return T


def _y_rows(y, B):
    """(y as (rows, n), sample stride) of a network output: a one-row ``y`` is SHARED by all B samples (stride 0) -- the output of a
    coupler network whose input is structurally zero is the same for every sample (``net_primal_zero_input``)."""
    y2 = y.view(y.shape[0], -1)
    assert y.shape[0] in (1, B)
    return y2, (0 if y.shape[0] == 1 and B > 1 else y2.shape[1])


def _t3(T):
    return (None, 0, 0) if T is None else (_p(T.data), T.t_b, T.t_r)


def acl_primal(z, y, maps, decode, lj=None):
    B = z.shape[0]
    z2 = z.view(B, -1)
    y2, y_b = _y_rows(y, B)
    _lib.check(_lib.load().cmf_acl_primal(_p(z2), z2.shape[1], _p(y2), y_b, _p(maps["zi"]), _p(maps["si"]),
                                          _p(maps["ti"]), maps["n"], B, int(decode), _p(lj), _stream()), "cmf_acl_primal")


def acl_tangent(T, YT, z, y, g, maps):
    """``YT`` None: the network's tangent is identically zero (structurally-zero network input): T_mod <- e^{-s} T_mod."""
    B = z.shape[0]
    z2 = z.view(B, -1)
    y2, y_b = _y_rows(y, B)
    launch = lambda: _lib.check(_lib.load().cmf_acl_tangent(_p(T.data), T.t_b, T.t_r, *_t3(YT), T.nc, _p(z2),
                                                            z2.shape[1], _p(y2), y_b, _p(g), _p(maps["zi"]), _p(maps["si"]),
                                                            _p(maps["ti"]), maps["n"], B, _stream()), "cmf_acl_tangent")
    TIMER = _timer()
    if TIMER is None:
        return launch()
    # coupling pass, per modified element: tangent rows v, s-dot, t-dot in (3 x NC floats), one row out, 5 scalars
    n = float(maps["n"]) * B
    TIMER.wrap("acl_tangent", 4.0 * n * T.nc, 4.0 * n * ((4 if YT is not None else 2) * T.nc + 5), launch)


def acl_cotangent(Ct, YC, z, y, g, maps):
    """Adjoint of acl_tangent on the cotangent tensor ``Ct`` (in place on the modified rows); fills the modified rows of
    ``YC`` (cotangent of the coupler network's raw output; the caller zeroed it; None: not wanted)."""
    B = z.shape[0]
    z2 = z.view(B, -1)
    y2, y_b = _y_rows(y, B)
    _lib.check(_lib.load().cmf_acl_cotangent(_p(Ct.data), Ct.t_b, Ct.t_r, *_t3(YC), Ct.nc, _p(z2),
                                             z2.shape[1], _p(y2), y_b, _p(g), _p(maps["zi"]), _p(maps["si"]),
                                             _p(maps["ti"]), maps["n"], B, _stream()), "cmf_acl_cotangent")


def modified_rows(T, maps):
    """Compact copy (B, n_mod, nc) of the rows of ``T`` a coupling layer is about to update (kept for acl_cross_terms)."""
    out = Tangent(T.B, maps["n"], T.nc, T.layout, T.data.device)
    _lib.check(_lib.load().cmf_gather_tangent(_p(T.data), T.t_b, T.t_r, _p(out.data), out.t_b, out.t_r, _p(maps["zi"]), maps["n"],
                                              T.nc, T.B, _stream()), "cmf_gather_tangent")
    return out


def acl_primal_backward(dx, z, y, maps, dy, decode, dlj=None):
    """Primal backward of the coupling update, in place on ``dx`` (z-shaped); accumulates into ``dy`` (y-shaped).  ``z`` = the
    tensor the forward update read (decode: before the update; encode: the layer input)."""
    B = z.shape[0]
    d2, z2, y2 = dx.view(B, -1), z.view(B, -1), y.view(B, -1)
    _lib.check(_lib.load().cmf_acl_primal_backward(_p(d2), d2.shape[1], _p(z2), z2.shape[1], _p(y2), y2.shape[1], _p(dy), _p(maps["zi"]),
                                                   _p(maps["si"]), _p(maps["ti"]), maps["n"], B, int(decode), _p(dlj), _stream()),
               "cmf_acl_primal_backward")


def acl_cross_terms(Ct, V, YT, z, y, g, maps, dz, dy, dg):
    """Primal cotangents of the tangent update (training): column reductions accumulated into ``dz`` (like z), ``dy`` (like y:
    the log-scale entries) and ``dg`` (like g).  ``Ct`` must still hold the cotangent of the UPDATED rows: call this before
    ``acl_cotangent``.  ``V`` = ``modified_rows(T, maps)`` taken before ``acl_tangent``, ``YT`` = the network's raw tangent."""
    B = z.shape[0]
    z2, y2 = z.view(B, -1), y.view(B, -1)
    # YT None (network tangent identically zero): only the log-scale entries of dy receive a term; dz / dg are not touched
    _lib.check(_lib.load().cmf_acl_cross_terms(_p(Ct.data), Ct.t_b, Ct.t_r, _p(V.data), V.t_b, V.t_r, *_t3(YT),
                                               Ct.nc, _p(z2), z2.shape[1], _p(y2), y2.shape[1], _p(g), _p(maps["zi"]),
                                               _p(maps["si"]), _p(maps["ti"]), maps["n"], B, _p(dz), _p(dy), _p(dg), _stream()),
               "cmf_acl_cross_terms")


def relu_bits(act):
    """BitMask of [act > 0] for an activation tensor (B, C, H, W) (CMF_F_RELU_BITS layout)."""
    B, Cc = act.shape[0], act.shape[1]
    HW = act[0, 0].numel()
    m = BitMask(B, HW, Cc, act.device)
    _lib.check(_lib.load().cmf_relu_bits(_p(act.contiguous()), _p(m.data), B, Cc, HW, _stream()), "cmf_relu_bits")
    return m


def accumulate(dst, src):
    """dst += src (flat fp32 tensors of equal size; any length / alignment: cmf_accumulate has a scalar sweep for odd ones)."""
    _lib.check(_lib.load().cmf_accumulate(_p(dst), _p(src), min(dst.numel(), src.numel()), _stream()), "cmf_accumulate")


def accumulate_any(dst, src):
    """dst += src for flat fp32 tensors of any length (kept as a name: allocates nothing since round 4)."""
    assert src.numel() == dst.numel()
    accumulate(dst, src)


def stanh_backward(dy, dg, y, g, sw, sb, dsw=None, dsb=None):
    """ScaledTanh output stage backward: returns du for y = sw tanh(u) + sb, g = sw (1 - tanh^2) given the cotangents of y and g
    (``dg`` may be None); accumulates the parameter gradients into ``dsw`` / ``dsb`` (C floats each) when given."""
    B, Cc = y.shape[0], y.shape[1]
    HW = y[0, 0].numel()
    du = torch.empty_like(y)
    _lib.check(_lib.load().cmf_stanh_backward(_p(dy.contiguous()), _p(None if dg is None else dg.contiguous()), _p(y), _p(g),
                                              _p(sw.detach().reshape(-1).contiguous()), _p(sb.detach().reshape(-1).contiguous()),
                                              _p(du), _p(dsw), _p(dsb), B, Cc, HW, _stream()), "cmf_stanh_backward")
    return du


def channel_sum(t, t_np, t_c, t_px, np_, Cc, npx, nc, out, t_sl=16):
    """out[c] += sum of a tangent-layout tensor over samples, pixels and columns (bias gradients)."""
    _lib.check(_lib.load().cmf_channel_sum(_p(t), int(t_np), int(t_c), int(t_px), int(t_sl), int(np_), int(Cc), int(npx), int(nc),
                                           _p(out), _stream()), "cmf_channel_sum")


def channel_sum_batched(ts, t_np, t_c, t_px, np_, Cc, npx, nc, outs, t_sl=16):
    """``channel_sum`` for up to 16 tensors of one shape in one launch (lists of tensors and of the bias-gradient vectors)."""
    n = len(ts)
    assert n == len(outs) and 1 <= n <= _lib.WGRAD_MAX_BATCH
    arr = lambda xs: (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    _lib.check(_lib.load().cmf_channel_sum_batched(arr(ts), arr(outs), n, int(t_np), int(t_c), int(t_px), int(t_sl), int(np_), int(Cc), int(npx),
                                                   int(nc), _stream()), "cmf_channel_sum_batched")


class GramResult:
    __slots__ = ("jtj", "logdet", "l1_off", "l1_diag", "info", "fail", "attempts")


def gram_cholesky(T, d, max_attempts=6, eps0=1e-6):
    """Fused Gram + Cholesky with the reference's whole-batch jitter retries enqueued back to back."""
    lib = _lib.load()
    dev = T.data.device
    r = GramResult()
    r.jtj = torch.empty(T.B, d, d, dtype=torch.float32, device=dev)
    r.logdet = torch.empty(T.B, dtype=torch.float32, device=dev)
    r.l1_off = torch.empty(T.B, dtype=torch.float32, device=dev)
    r.l1_diag = torch.empty(T.B, dtype=torch.float32, device=dev)
    r.info = torch.empty(T.B, dtype=torch.int32, device=dev)
    r.fail = torch.empty(8, dtype=torch.int32, device=dev)
    launch = lambda: _lib.check(lib.cmf_gram_cholesky(_p(T.data), T.t_b, T.t_r, T.N, T.nc, d, T.B, _p(r.jtj), _p(r.logdet),
                                                      _p(r.l1_off), _p(r.l1_diag), _p(r.info), _p(r.fail), _stream()),
                                "cmf_gram_cholesky")
    TIMER = _timer()
    if TIMER is None:
        launch()
    else:                                          # 2 D d^2 (+ d^3/3) FLOP and (D NC + d^2 + 3) 4 B per sample
        TIMER.wrap("gram_cholesky", T.B * (2.0 * T.N * d * d + d ** 3 / 3.0), 4.0 * T.B * (T.N * T.nc + d * d + 3), launch)
    cholesky_retries(r, d, max_attempts, eps0)
    return r


def cholesky_retries(r, d, max_attempts=6, eps0=1e-6):
    """Enqueue the whole-batch jitter retries 1 .. max_attempts-1 behind a factorisation (non_square.py:280-288): retry ``a``
    exits at once unless ``fail[a-1]`` is set, else adds eps0 * 10^(a-1) to every sample's diagonal IN PLACE and factorises again."""
    lib = _lib.load()
    B = r.jtj.shape[0]
    for a in range(1, max_attempts):
        _lib.check(lib.cmf_cholesky_retry(_p(r.jtj), d, B, a, eps0, _p(r.logdet), _p(r.l1_diag), _p(r.info), _p(r.fail),
                                          _stream()), "cmf_cholesky_retry")


def gram_backward(T, jtj, g_logdet=None, g_l1off=None, g_l1diag=None):
    """Cotangent of the Jacobian stack T for a loss with d/d logdet = g_logdet, d/d l1_off = g_l1off, d/d l1_diag =
    g_l1diag (each (B,) or None): what autograd yields through non_square.py:307-308, :280-294, :87-100."""
    B, d = jtj.shape[0], jtj.shape[1]
    assert B == T.B and d <= T.nc
    dT = T.like(T.N)
    gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_logdet, g_l1off, g_l1diag)]
    _lib.check(_lib.load().cmf_gram_backward(_p(T.data), T.t_b, T.t_r, T.N, T.nc, d, B, _p(jtj),
                                             *[None if g is None else _p(g) for g in gs], _p(dT.data), dT.t_b, dT.t_r,
                                             _stream()), "cmf_gram_backward")
    return dT


def gram_backward_matrix(T, M):
    """dT = T (M + M^T) for an explicit cotangent M (B, d, d) of the Gram matrix (Hutchinson surrogate training)."""
    B, d = M.shape[0], M.shape[1]
    assert B == T.B and d <= T.nc
    dT = T.like(T.N)
    _lib.check(_lib.load().cmf_gram_backward_matrix(_p(T.data), T.t_b, T.t_r, T.N, T.nc, d, B, _p(M.to(torch.float32).contiguous()),
                                                    _p(dT.data), dT.t_b, dT.t_r, _stream()), "cmf_gram_backward_matrix")
    return dT


def hutch_cg(jtj, eps, max_iter, tol, min_iter=None):
    """Hutchinson surrogate on explicit J^T J: returns (value (B,), u, w (B,d,S), iterations (B,))."""
    B, d, S = eps.shape
    if min_iter is None:
        min_iter = min(10, max_iter - 1) + 1 if max_iter > 1 else 1
    dev = eps.device
    u, w = torch.empty_like(eps), torch.empty_like(eps)
    val = torch.empty(B, dtype=torch.float32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().cmf_hutch_cg(_p(jtj), _p(eps.contiguous()), d, S, B, int(max_iter), int(min_iter), float(tol),
                                        _p(u), _p(w), _p(val), _p(iters), _stream()), "cmf_hutch_cg")
    return val, u, w, iters


def hutch_metric(w):
    """(l1_off, l1_diag) of the Hutchinson product W = (J^T J) eps, (B, d, S) (non_square.py:87-100 on :253-258).  The
    off-diagonal sum exists only for S == d (the reference's ``view`` at :98; None otherwise); the diagonal one is
    ``torch.diagonal`` of the rectangular block: min(d, S) entries (:87-92)."""
    B, d, S = w.shape
    off = torch.empty(B, dtype=torch.float32, device=w.device) if S == d else None
    diag = torch.empty(B, dtype=torch.float32, device=w.device)
    _lib.check(_lib.load().cmf_hutch_metric(_p(w.contiguous()), d, S, B, _p(off), _p(diag), _stream()), "cmf_hutch_metric")
    return off, diag


def hutch_cotangent(u, eps, w, g_val=None, g_off=None, g_diag=None):
    """d objective / d (J^T J) as an explicit (B, d, d) matrix for the train-mode Hutchinson objective (u detached)."""
    B, d, S = eps.shape
    M = torch.empty(B, d, d, dtype=torch.float32, device=eps.device)
    gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_val, g_off, g_diag)]
    _lib.check(_lib.load().cmf_hutch_cotangent(_p(u.contiguous()), _p(eps.contiguous()), _p(w.contiguous()), d, S, B,
                                               *[None if g is None else _p(g) for g in gs], _p(M), _stream()), "cmf_hutch_cotangent")
    return M


def hutch_lowrank_cotangent(w, S, g_val=None, g_diag=None):
    """(B, n, n) cotangent of P^T P for the n-column sweep P = J [u | eps | e_0..e_{K-1}] of the low-rank Hutchinson backward
    (``cmf_hutch_lowrank_cotangent``); n = 2 S + (min(d, S) if g_diag is given)."""
    B, d = w.shape[0], w.shape[1]
    n = 2 * S + (min(d, S) if g_diag is not None else 0)
    M = torch.empty(B, n, n, dtype=torch.float32, device=w.device)
    gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_val, g_diag)]
    _lib.check(_lib.load().cmf_hutch_lowrank_cotangent(_p(w.contiguous()), d, S, B, *[None if g is None else _p(g) for g in gs], n,
                                                       _p(M), _stream()), "cmf_hutch_lowrank_cotangent")
    return M


# --------------------------------------------------------------------------------------------------
# NSF prior layers: derived weights are cached per parameter version like the packs
# --------------------------------------------------------------------------------------------------


class _DerivedCache:
    """Tensors computed from parameters by a kernel (masked MADE weights, L U products), rebuilt when a source parameter
    changes (``_version`` / storage / ``PACKS.generation``, which FlatOptimizer bumps).  A stale entry is rebuilt INTO its old
    tensors (same identity, ``_version`` bumped by the copy), so the pack cache entry keyed on that identity is refreshed in place
    instead of a new one piling up per optimiser step."""

    def __init__(self):
        self._store = {}

    def get(self, key, sources, build):
        import weakref
        ver = tuple((s._version, s.data_ptr(), s.device) for s in sources) + (PACKS.generation,)
        hit = self._store.get(key)
        if hit is not None and hit[0]() is sources[0] and hit[1] == ver:
            return hit[2]
        out = build()
        if hit is not None and hit[0]() is sources[0]:
            # same layout as the stale entry (equal shapes / dtypes; equal non-tensor parts such as offsets and widths): refresh
            # the OLD tensors in place, so whatever is keyed on their identity (the pack cache) follows instead of piling up
            olds, news = (hit[2], out) if isinstance(out, tuple) else ((hit[2],), (out,))
            same = len(olds) == len(news) and all(
                (o.shape == n.shape and o.dtype == n.dtype and o.device == n.device) if torch.is_tensor(o) and torch.is_tensor(n)
                else (not torch.is_tensor(o) and not torch.is_tensor(n) and o == n) for o, n in zip(olds, news))
            if same:
                for o, n in zip(olds, news):
                    if torch.is_tensor(o):
                        o.copy_(n)
                out = hit[2]
        if len(self._store) > 4096:
            self._store = {k: v for k, v in self._store.items() if v[0]() is not None}
        self._store[key] = (weakref.ref(sources[0]), ver, out)
        return out


DERIVED = _DerivedCache()


#: fused coupling layers for MLP couplers (cmf_mlp_coupler): False = always the per-layer launches (the training path
#: always uses them: it needs every layer's state)
FUSED_MLP = True


def mlp_coupler_supported(net, view, T, n_out):
    """Shapes the fused kernel covers: tanh MLP, <= 8 linear layers, hidden <= 128, outputs <= 64, inputs <= 128; with
    tangents: 16 column slots of which at most 15 are Jacobian columns (column 15 carries the primal)."""
    if not FUSED_MLP or net.kind != "mlp" or view.mask is not None:
        return False
    lins = [m for m in net if isinstance(m, nn.Linear)]
    if not 2 <= len(lins) <= _lib.MLP_MAX_LAYERS or view.cin > 128 or n_out > 64:
        return False
    if any(l.out_features > 128 for l in lins[:-1]):
        return False
    return T is None or (T.layout == "fmajor" and T.nc == 16)


def _mlp_images(net):
    """All layer images of an MLP coupler net back to back (cmf_pack_mlp_layer), cached per parameter version:
    (flat tensor, float offsets, widths)."""
    lins = [m for m in net if isinstance(m, nn.Linear)]
    params = [p for l in lins for p in (l.weight, l.bias)]

    def build():
        lib = _lib.load()
        ht = lib.cmf_mlp_hidden_tiles(max(l.out_features for l in lins[:-1]))
        tiles = [(ht if i + 1 < len(lins) else (l.out_features + 15) // 16, ht if i > 0 else (l.in_features + 15) // 16)
                 for i, l in enumerate(lins)]                    # (output tiles, input K groups) per layer
        sizes = []
        for l, (mt, kg) in zip(lins, tiles):
            n = C.c_longlong(0)
            _lib.check(lib.cmf_pack_mlp_layer(None, None, l.out_features, l.in_features, 0, mt, kg, None, C.byref(n), None), "pack size")
            sizes.append((n.value + 3) // 4 * 4)
        offs = [0]
        for n in sizes[:-1]:
            offs.append(offs[-1] + n)
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=lins[0].weight.device)
        for i, (l, (mt, kg)) in enumerate(zip(lins, tiles)):
            _lib.check(lib.cmf_pack_mlp_layer(_p(l.weight.detach().contiguous()), _p(l.bias.detach().contiguous()), l.out_features,
                                              l.in_features, int(i == 0), mt, kg, C.c_void_p(flat.data_ptr() + 4 * offs[i]), None,
                                              _stream()), "cmf_pack_mlp_layer")
        return flat, offs, [lins[0].in_features] + [l.out_features for l in lins]

    return DERIVED.get((id(lins[0].weight), "mlp-images"), params, build)


def mlp_coupler(net, z, T, view, maps, decode, lj=None, ncols=None):
    """One fused coupling layer with an MLP coupler: in place on ``z`` (B, D) and, with a tangent stack ``T`` (fmajor, 16
    columns of which ``ncols`` <= 15 are used), on the modified rows of ``T``."""
    flat, offs, widths = _mlp_images(net)
    B = z.shape[0]
    z2 = z.view(B, -1)
    a = _lib.MlpCouplerArgs()
    a.z, a.z_b = _p(z2), z2.shape[1]
    if T is not None:
        assert decode and T.layout == "fmajor" and T.nc == 16 and (ncols is None or ncols <= 15)
        a.t, a.t_f = _p(T.data), T.t_r
    a.w = _p(flat)
    a.zi, a.si, a.ti, a.n_mod = _p(maps["zi"]), _p(maps["si"]), _p(maps["ti"]), maps["n"]
    a.B, a.cin, a.chan_off, a.chan_step = B, view.cin, view.chan_off, view.chan_step
    a.n_layers = len(offs)
    for i, w_ in enumerate(widths):
        a.width[i] = w_
    for i, o in enumerate(offs):
        a.w_off[i] = o
    a.decode = int(decode)
    a.lj = _p(lj)
    a.ncols = int(ncols) if (T is not None and ncols is not None) else 15
    launch = lambda: _lib.check(_lib.load().cmf_mlp_coupler(C.byref(a), _stream()), "cmf_mlp_coupler")
    TIMER = _timer()
    if TIMER is None:
        return launch()
    # algorithmic work: 2 in out FLOP per layer and column (1 primal + ncols tangents, or 1 per sample in primal mode)
    cols = (1 + (ncols if ncols is not None else 15)) if T is not None else 1
    fl = 2.0 * B * cols * sum(widths[i] * widths[i + 1] for i in range(len(widths) - 1))
    by = 4.0 * B * cols * (view.cin + 2 * maps["n"])
    TIMER.wrap("mlp_coupler" + ("_tangent" if T is not None else "_primal"), fl, by, launch)


def made_masked_weight(weight, kind, features, multiplier=1):
    """weight * MADE mask (kind 0 input->hidden, 1 hidden->hidden, 2 hidden->output) as a cached (out, in) tensor."""
    def build():
        w = weight.detach().contiguous()
        out = torch.empty_like(w)
        _lib.check(_lib.load().cmf_made_mask_weight(_p(w), _p(out), w.shape[0], w.shape[1], int(kind), int(features),
                                                    int(multiplier), _stream()), "cmf_made_mask_weight")
        return out
    return DERIVED.get((id(weight), "made", kind, features, multiplier), [weight], build)


def lu_weights(lower, upper, udiag, eps=1e-3):
    """(W = L U (n, n), logabsdet (1,)) of an LULinear layer, cached per parameter version."""
    def build():
        n = udiag.shape[0]
        W = torch.empty(n, n, dtype=torch.float32, device=udiag.device)
        ld = torch.empty(1, dtype=torch.float32, device=udiag.device)
        _lib.check(_lib.load().cmf_lu_weights(_p(lower.detach().contiguous()), _p(upper.detach().contiguous()),
                                              _p(udiag.detach().contiguous()), n, float(eps), _p(W), _p(ld), _stream()),
                   "cmf_lu_weights")
        return W, ld
    return DERIVED.get((id(udiag), "lu"), [udiag, lower, upper], build)


def linear_primal(x, weight, bias, relu_in=False, res=None):
    """y (B, out) = W [relu](x) + b [+ res] through ``cmf_conv_primal`` with taps = 1 (pixels = samples)."""
    B, cin = x.shape
    cout = weight.shape[0]
    y = torch.empty(B, cout, dtype=torch.float32, device=x.device)
    conv_primal(x, 0, 0, 1, cin, weight, 1, bias, y, 0, 1, cout, 1, cin, cout, 1, B, imode=F_RELU if relu_in else F_NONE, res=res)
    return y


class GroupedBatch:
    """(B, F) tensors with 16 samples in the 16 column slots of the tangent-conv kernels -- (G, F, 16), viewed as ONE image row of
    G pixels -- for the backward of small linear layers (MLP couplers, the nsf prior's MADE and LULinear) on those kernels."""

    def __init__(self, B, device):
        self.B, self.Bp, self.dev = B, (B + 15) // 16 * 16, device
        self.G = self.Bp // 16

    def pack(self, t):
        t = t.reshape(self.B, -1)
        if self.Bp != self.B:
            t = torch.cat([t, torch.zeros(self.Bp - self.B, t.shape[1], dtype=t.dtype, device=self.dev)])
        return primal_regroup(t.contiguous(), True)

    def unpack(self, t_g, F):
        return primal_regroup(t_g.view(self.G, -1), False).view(self.Bp, F)[: self.B]

    @staticmethod
    def strides(c):
        return (0, 16, c * 16)                               # (np, chan, px)

    def linear_backward(self, x_g, dy_g, weight, cin, cout, dw=None, db=None, relu_in=False, fo_g=None, res_g=None):
        """y = W [relu](x) + b on grouped tensors: dw += sum dy (x) [relu](x), db += sum dy, returns W^T dy (optionally
        times [fo > 0], plus res)."""
        if dw is not None:
            conv_tangent_wgrad(x_g, 0, *self.strides(cin), dy_g, 0, *self.strides(cout), dw, 1, 1, cin, cout, 1, self.G, 16,
                               fmode=F_SELF_RELU if relu_in else F_NONE)
        if db is not None:
            channel_sum(dy_g, *self.strides(cout), 1, cout, self.G, 16, db)
        dx_g = torch.empty(self.G * cin * 16, dtype=torch.float32, device=self.dev)
        fo = {} if fo_g is None else dict(fo=fo_g, fo_np=0, fo_co=16, fo_px=cin * 16, fomode=F_SELF_RELU)
        conv_tangent(dy_g, 0, *self.strides(cout), weight, 1, dx_g, *self.strides(cin), 1, cout, cin, 1, self.G, 16,
                     transpose=True, precision="f32", res_t=res_g, **fo)
        return dx_g


def rq_spline_backward(x, params, bins, hidden, tail_bound, dz, dlj=None):
    """(dx (B, D), dparams (B, D * (3 bins - 1))) of the forward spline for the cotangents dz (B, D) and dlj (B,)."""
    B, D = x.shape
    dx = torch.empty_like(x)
    dparams = torch.empty_like(params)
    _lib.check(_lib.load().cmf_rq_spline_backward(_p(x), D, _p(params), D, int(bins), int(hidden), float(tail_bound), B,
                                                  _p(dz.contiguous()), D, _p(dlj), _p(dx), D, _p(dparams), _stream()),
               "cmf_rq_spline_backward")
    return dx, dparams


def rq_spline(x, params, bins, hidden, tail_bound, inverse=False, lj=None):
    """Elementwise rational-quadratic spline with linear tails on (B, D); params (B, D * (3 bins - 1)); lj (B,) accumulates."""
    B, D = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.load().cmf_rq_spline(_p(x), D, _p(params), D, int(bins), int(hidden), float(tail_bound), int(inverse), B,
                                         _p(out), D, _p(lj), _stream()), "cmf_rq_spline")
    return out


def prehead(x, noise, a, c, logit):
    B = x.shape[0]
    n = x[0].numel()
    y = torch.empty_like(x)
    lj = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().cmf_prehead(_p(x), _p(noise), _p(y), _p(lj), float(a), float(c), int(logit), n, B, _stream()),
               "cmf_prehead")
    return y, lj


def prehead_inverse(y, a, c, logit):
    x = torch.empty_like(y)
    _lib.check(_lib.load().cmf_prehead_inverse(_p(y), _p(x), float(a), float(c), int(logit), y.numel(), _stream()),
               "cmf_prehead_inverse")
    return x


def gaussian_logprob(z, lp):
    B = z.shape[0]
    z2 = z.view(B, -1)
    _lib.check(_lib.load().cmf_gaussian_logprob(_p(z2), z2.shape[1], z2.shape[1], B, _p(lp), _stream()),
               "cmf_gaussian_logprob")


def affine_prior(z, log_scale, shift, decode, lj=None):
    B = z.shape[0]
    z2 = z.view(B, -1)
    _lib.check(_lib.load().cmf_affine_prior(_p(z2), z2.shape[1], _p(log_scale.detach().contiguous()),
                                            _p(shift.detach().contiguous()), z2.shape[1], B, int(decode), _p(lj), _stream()),
               "cmf_affine_prior")


def recon_sqerr(xh, x):
    B = x.shape[0]
    rec = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().cmf_recon_sqerr(_p(xh.reshape(B, -1)), _p(x.reshape(B, -1)), x[0].numel(), B, _p(rec), _stream()),
               "cmf_recon_sqerr")
    return rec


def elbo_combine(low, logdet, rec, l1, pre, wl, lam, wm, B, device):
    out = torch.empty(B, 1, dtype=torch.float32, device=device)
    _lib.check(_lib.load().cmf_elbo_combine(_p(low), _p(logdet), _p(rec), _p(l1), _p(pre), float(wl), float(lam), float(wm),
                                            B, _p(out), _stream()), "cmf_elbo_combine")
    return out


# --------------------------------------------------------------------------------------------------
# coupler networks
# --------------------------------------------------------------------------------------------------


class NetView:
    """Where a coupler network reads its input inside the current primal / tangent tensors."""

    def __init__(self, geom, cin, chan_off=0, chan_step=1, mask=None, live=None):
        self.geom, self.cin, self.chan_off, self.chan_step, self.mask = geom, cin, chan_off, chan_step, mask
        #: checkerboard couplers: {"parity": 1 | 2, "act_idx": int32 map (hid * HW/2,) gathering the live pixels of a (hid, H, W)
        #: activation} -- the network's output is read at the pixels with (row + col) % 2 == parity - 1 only (net_tangent)
        self.live = live


class Geometry:
    """Shape of the tensor an ACL acts on: image (C, H, W) or flat (F,)."""

    def __init__(self, shape):
        self.shape = tuple(int(s) for s in shape)
        self.image = len(self.shape) == 3
        if self.image:
            self.C, self.H, self.W = self.shape
            self.HW = self.H * self.W
        else:
            assert len(self.shape) == 1
            self.C, self.H, self.W, self.HW = self.shape[0], 1, 1, 1
        self.N = int(np.prod(self.shape))


def _resnet_parts(net):
    body = net.module
    blocks = [m for m in body if hasattr(m, "conv1")]
    convs = [m for m in body if isinstance(m, nn.Conv2d)]
    return convs[0], blocks, convs[-1]


def net_primal(net, z, view, need_acts=True):
    """Primal forward of a coupler network on the current tensor ``z`` (B, *geom.shape).
    Returns (y (B, cout, ...), g or None, acts: activation tensors the tangent pass differentiates through;
    ``need_acts=False`` (encode pass, sampling) skips preparing them)."""
    geo, B, dev = view.geom, z.shape[0], z.device
    if net.kind == "resnet":
        conv0, blocks, convf = _resnet_parts(net)
        hid, cout, H, W, HW = conv0.out_channels, convf.out_channels, geo.H, geo.W, geo.HW
        new = lambda c: torch.empty(B, c, H, W, dtype=torch.float32, device=dev)
        a = new(hid)
        conv_primal(z, view.chan_off * HW, geo.C * HW, view.chan_step * HW, 1, conv0.weight, 9, None, a, hid * HW, HW, 1,
                    B, view.cin, hid, H, W, imode=F_RAW if view.mask is not None else F_NONE, mask=view.mask, f_c=HW, f_px=1)
        if B % 16 == 0 and _shape_ok_bf16x3(9, hid, W, False, H, hid):
            return _resnet_primal_grouped(net, blocks, convf, a, B, hid, cout, H, W, need_acts)
        need_acts = True if need_acts == "train" else need_acts
        acts = [a]
        for blk in blocks:
            c1 = new(hid)
            conv_primal(a, 0, hid * HW, HW, 1, blk.conv1.weight, 9, blk.conv1.bias, c1, hid * HW, HW, 1, B, hid, hid, H, W,
                        imode=F_RELU)
            a2 = new(hid)
            conv_primal(c1, 0, hid * HW, HW, 1, blk.conv2.weight, 9, blk.conv2.bias, a2, hid * HW, HW, 1, B, hid, hid, H, W,
                        imode=F_RELU, res=a)
            acts += [c1, a2]
            a = a2
        y, g = new(cout), new(cout)
        conv_primal(a, 0, hid * HW, HW, 1, convf.weight, 1, convf.bias, y, cout * HW, HW, 1, B, hid, cout, H, W,
                    imode=F_RELU, omode=O_STANH, sw=net.weights.detach().reshape(-1), sb=net.bias.detach().reshape(-1), g=g)
        return y, g, acts
    # MLP: pixels = batch samples, channels = features
    lins = [m for m in net if isinstance(m, nn.Linear)]
    F_in = geo.C
    h, h_c, h_px, h_off, cin = z, view.chan_step, F_in, view.chan_off, view.cin
    acts = []
    for i, lin in enumerate(lins):
        last = i == len(lins) - 1
        out = torch.empty(B, lin.out_features, dtype=torch.float32, device=dev)
        conv_primal(h, h_off, 0, h_c, h_px, lin.weight, 1, lin.bias, out, 0, 1, lin.out_features, 1, cin, lin.out_features,
                    1, B, omode=O_NONE if last else O_TANH)
        if not last:
            acts.append(out)
        h, h_c, h_px, h_off, cin = out, 1, lin.out_features, 0, lin.out_features
    return h, None, acts


def net_primal_zero_input(net, view, B, device):
    """(y, g), one row each, of a coupler network whose input is structurally ZERO -- the channels ``SplitDensity.pad_inputs``
    appended (split.py:50-52) feeding a coupler that passes exactly those through (acl.py:148-160): the output is the same for
    every sample, so ONE sample group runs (16 zero samples where the batch takes the sample-grouped kernels, 1 otherwise: the
    same kernels, hence the same bits, as the full batch) and the coupling kernels read it with sample stride 0.  Cached per
    parameter version and KernelConfig."""
    rows = 16 if B % 16 == 0 else 1
    params = list(net.parameters())

    def build():
        z0 = torch.zeros(rows, *view.geom.shape, dtype=torch.float32, device=device)
        y, g, _ = net_primal(net, z0, view, need_acts=False)
        return y[:1].clone(), (None if g is None else g[:1].clone())

    return DERIVED.get((id(params[0]), "zero-input", rows, cfg().primal, str(device)), params, build)


class GroupedActs(list):
    """Primal activations kept as (B/16, C, H, W, 16): 16 samples in the 16 column slots of the tangent kernels
    (supported as a factor source through ``f_group``, but slower to read than the standard layout)."""
    f_group = 16


class BitMask:
    """relu' of a hidden activation as one bit per (sample, pixel, channel): ``data`` (B, HW, C/8) uint8, the
    CMF_F_RELU_BITS layout (include/cmf_amd.h).  Written by the primal conv that produces the activation."""

    def __init__(self, B, HW, C, device):
        self.data = torch.empty(B, HW, C // 8, dtype=torch.uint8, device=device)
        self.np_bytes = HW * (C // 8)


class ActList(list):
    """Activations of a ResNet coupler kept for TRAINING: the elements are what the tangent pass and the reverse sweep read (the
    float a_0, relu' BitMasks of the hidden activations, the last activation as floats -- the "bits" form), and ``grouped`` holds
    ALL of them as the sample-grouped float tensors (B/16, C, HW, 16) the primal pass produced, which is what
    ``net_primal_backward`` reads.  (Before: 17 activations per coupler regrouped to the per-sample layout after the forward pass
    and back again in the backward pass -- 850 launches per training step.)"""
    grouped = None


def train_acts_mode(net, view, B, T=None, nc=None):
    """``need_acts`` for a training forward: "train" (ActList) where every consumer can work from bit masks and grouped floats,
    else True (per-sample float activations).  ``T`` = the tangent stack of the sweep that will read the activations, or ``nc``
    = its column slots when the sweep comes later (primal-only pass)."""
    nc = T.nc if T is not None else nc
    if net.kind != "resnet" or B % 32 or cfg().primal == "bf16x3":        # (the bf16 split kernel writes no bit masks)
        return True
    conv0 = _resnet_parts(net)[0]
    hid, H, W = conv0.out_channels, view.geom.H, view.geom.W
    if hid % 64 or not _shape_ok_bf16x3(9, hid, W, True, H, hid):
        return True
    if nc is not None and (cfg().tangent != "bf16x3" or nc % 32):
        return True
    return "train"


def _resnet_primal_grouped(net, blocks, convf, a0, B, hid, cout, H, W, need_acts):
    """Hidden 3x3 convs of the primal ResNet through the TANGENT conv kernels: the 16 column slots carry 16 samples (relu
    applied elementwise on load, bias as per-channel constant); ``cfg().primal`` picks the kernel (KernelConfig).
    need_acts: False (encode pass, sampling), True (float activations in the standard layout: fp32 tangent path, reverse
    sweep) or "bits" (the split-precision tangent pass follows: the fp32 kernel writes relu' bit masks next to each
    activation, 1/32 of the bytes, and nothing is regrouped but the last activation, which the 1x1 conv reads as floats)."""
    HW, G, dev = H * W, B // 16, a0.device
    pn = (hid * HW * 16, HW * 16, 16)                       # (np, chan, px) strides of a grouped tensor
    new = lambda: torch.empty(G * hid * HW * 16, dtype=torch.float32, device=dev)
    train = need_acts == "train"                          # ActList: the "bits" form + the grouped floats
    prec = cfg().primal
    if (prec == "f16x3" and hid % 64) or (prec in ("f16x3", "bf16x3") and hid > 64):
        # the split kernels work on whole 64-channel groups and fetch the bias of ONE group per launch (cmf_conv_tangent_f16x3
        # returns CMF_EINVAL for a bias with cout > 64): wider networks (g_hidden_channels 128, 192, ...) take the fp32 kernel
        prec = "f32"
    bits = (need_acts == "bits" or train) and prec != "bf16x3" and hid % 16 == 0
    a = primal_regroup(a0, True)
    acts, masks = [a], []
    # fp16 split: the input-range chain.  Word i = max |activation i| (word 0 by a reduction over a_0, the others raised by the
    # conv that stores the activation); a conv scales its input by the power of two that puts that maximum in [2^13, 2^14)
    # before the fp16 split.  A residual block's output is read by the next block's conv1 only, its conv1 output by conv2 only.
    rng = None
    if prec == "f16x3":
        rng = torch.zeros(2 * len(blocks) + 1, dtype=torch.float32, device=dev)
        absmax(a, rng[0:1])
    ax = lambda i: dict(amax_in=rng[i:i + 1], amax_out=rng[i + 1:i + 2]) if rng is not None else {}
    for k, blk in enumerate(blocks):
        c1, a2 = new(), new()
        m1, m2 = (BitMask(B, HW, hid, dev), BitMask(B, HW, hid, dev)) if bits else (None, None)
        mo = lambda m: dict(mask_out=m.data, mask_np=m.np_bytes) if m is not None else {}
        conv_tangent(a, 0, *pn, blk.conv1.weight, 9, c1, *pn, G, hid, hid, H, W, 16, fmode=F_SELF_RELU, bias=blk.conv1.bias,
                     precision=prec, **mo(m1), **ax(2 * k))
        conv_tangent(c1, 0, *pn, blk.conv2.weight, 9, a2, *pn, G, hid, hid, H, W, 16, fmode=F_SELF_RELU, bias=blk.conv2.bias,
                     res_t=a, precision=prec, **mo(m2), **ax(2 * k + 1))
        acts += [c1, a2]
        masks += [m1, m2]
        a = a2
    # 1x1 + ScaledTanh on the grouped tensor: a 1x1 conv does not care that "pixels" are (pixel, sample) pairs
    yg = torch.empty(G * cout * HW * 16, dtype=torch.float32, device=dev)
    gg = torch.empty_like(yg)
    conv_primal(a, 0, hid * HW * 16, HW * 16, 1, convf.weight, 1, convf.bias, yg, cout * HW * 16, HW * 16, 1, G, hid, cout,
                1, HW * 16, imode=F_RELU, omode=O_STANH, sw=net.weights.detach().reshape(-1), sb=net.bias.detach().reshape(-1),
                g=gg)
    y = primal_regroup(yg.view(G, -1), False).view(B, cout, H, W)
    g = primal_regroup(gg.view(G, -1), False).view(B, cout, H, W)
    if not need_acts:
        return y, g, None
    std_of = lambda t: primal_regroup(t.view(G, -1), False).view(B, hid, H, W)
    if train:
        out = ActList([a0] + masks[:-1] + [std_of(acts[-1])] if bits else [a0] + [std_of(t) for t in acts[1:]])
        out.grouped = acts
        return y, g, out
    if bits:
        return y, g, [a0] + masks[:-1] + [std_of(acts[-1])]
    # The tangent pass reads relu' per (channel, pixel) of ONE sample: from the grouped layout every such read is its
    # own 64-byte line (measured: tangent convs 3.1 -> 4.5 ms); regrouping the 17 saved activations costs ~6 ms / elbo.
    return y, g, [a0] + [std_of(t) for t in acts[1:]]


def _view_rows(T, view):
    """Compact copy of the rows of ``T`` a coupler network reads (``view``): cin channels, in T's layout."""
    if T.layout == "panel":
        HW = view.geom.HW
        rows = T.data[: T.B * T.N * T.nc].view(T.B, T.N // HW, HW * T.nc)[:, view.chan_off::view.chan_step][:, :view.cin]
        out = Tangent(T.B, view.cin * HW, T.nc, "panel", T.data.device)
    else:
        rows = T.data[: T.B * T.N * T.nc].view(T.N, T.B * T.nc)[view.chan_off::view.chan_step][:view.cin]
        out = Tangent(T.B, view.cin, T.nc, "fmajor", T.data.device)
    out.data.view(rows.shape).copy_(rows)
    return out


#: checkerboard couplers: last hidden conv and 1x1 conv at the (1 - mask) pixels only (False: every pixel, rounds 1 - 4)
CHECKERBOARD_TAIL = True


def net_tangent(net, T, view, acts, transpose_packs=False, save=None):
    """Push all Jacobian columns of ``T`` through the coupler network; returns the raw tangent of the
    network's pre-activation output (the ScaledTanh derivative ``g`` is applied by acl_tangent).
    ``save`` (a list, training): the input tangent of every conv / linear layer is kept and appended, in forward order
    (the network's own input as a compact copy of the rows it reads), for ``net_cotangent(..., saved=, grads=)``.
    Checkerboard couplers (``view.live``, evaluation only): a checkerboard layer reads (s, t, s-dot, t-dot) at its (1 - mask)
    pixels alone (acl.py:48-66) and the network's last layer is a pointwise 1x1 conv, so the LAST hidden conv (with its residual
    read) and the 1x1 conv run on those pixels only; the returned stack is then COMPACT -- (B, cout * HW/2, nc), pixel (row, col)
    at row * W/2 + col // 2 -- and carries ``compact = True`` (``AffineCouplingBijection.decode_`` switches index maps)."""
    geo, B, nc, dev = view.geom, T.B, T.nc, T.data.device
    if save is not None:
        save.append(_view_rows(T, view))
    if net.kind == "resnet":
        conv0, blocks, convf = _resnet_parts(net)
        hid, cout, H, W, HW = conv0.out_channels, convf.out_channels, geo.H, geo.W, geo.HW
        new = lambda c: Tangent(B, c * HW, nc, "panel", dev)
        pn = lambda c: (c * HW * nc, HW * nc, nc)          # (np, chan, px) strides of a panel with c channels
        # Hidden tangents live SLICE-MAJOR, [sample][pixel][16-column slice][channel][16]: the 16 channels x 16 columns an
        # MFMA wave stores (or reads as residual) per instruction are then one contiguous KiB instead of sixteen 64-byte
        # pieces in sixteen channel planes -- that access shape ran at 13 B/clk/CU against 48 (tools/ubench/curate2.py)
        # and was the 15k-cycle tail of every work item.  Only the kernels' strides know about it.
        hd = (hid * HW * nc, 16, hid * nc)                   # (np, chan, px)
        hsl = hid * 16                                       # slice stride
        h = new(hid)
        conv_tangent(T.data, view.chan_off * HW * nc, T.t_b, view.chan_step * HW * nc, nc, conv0.weight, 9, h.data, *hd,
                     B, view.cin, hid, H, W, nc, fmode=F_RAW if view.mask is not None else F_NONE, f=view.mask, f_np=0,
                     f_ci=HW, f_px=1, y_sl=hsl)
        fg = getattr(acts, "f_group", 1)                     # primal activations: (B,C,H,W) or (B/16,C,H,W,16)
        fs = dict(f_np=hid * HW * fg, f_ci=HW * fg, f_px=fg, f_group=fg)
        # relu' source of a hidden conv: float activations, or the bit mask the primal pass wrote (BitMask)
        fk = lambda t: dict(fmode=F_RELU_BITS, f=t.data, f_np=t.np_bytes) if isinstance(t, BitMask) else dict(fmode=F_RELU, f=t, **fs)
        u, h2 = new(hid), new(hid)
        compact = (CHECKERBOARD_TAIL and view.live is not None and save is None and fg == 1 and len(blocks) > 0 and hid % 64 == 0
                   and W % 2 == 0 and _use_bf16x3(9, hid, W, False, H, hid))
        for k, blk in enumerate(blocks):
            a_in, c1 = acts[2 * k], acts[2 * k + 1]
            conv_tangent(h.data, 0, *hd, blk.conv1.weight, 9, u.data, *hd, B, hid, hid, H, W, nc, x_sl=hsl, y_sl=hsl, **fk(a_in))
            if compact and k + 1 == len(blocks):
                # the last hidden conv at the live pixels only, stored compactly; its residual read at those pixels of the full h
                HWc = HW // 2
                hc = Tangent(B, hid * HWc, nc, "panel", dev)
                conv_tangent(u.data, 0, *hd, blk.conv2.weight, 9, hc.data, hid * HWc * nc, 16, hid * nc, B, hid, hid, H, W, nc,
                             res_t=h.data, res_np=hd[0], x_sl=hsl, y_sl=hsl, live=view.live["parity"], **fk(c1))
                a_last = gather_primal(acts[-1], view.live["act_idx"], hid * HWc)         # relu' source of the 1x1 conv, compact too
                yt = Tangent(B, cout * HWc, nc, "panel", dev)
                conv_tangent(hc.data, 0, hid * HWc * nc, 16, hid * nc, convf.weight, 1, yt.data, cout * HWc * nc, HWc * nc, nc, B, hid,
                             cout, H, W // 2, nc, fmode=F_RELU, f=a_last, x_sl=hsl, f_np=hid * HWc, f_ci=HWc, f_px=1)
                yt.compact = True
                return yt
            conv_tangent(u.data, 0, *hd, blk.conv2.weight, 9, h2.data, *hd, B, hid, hid, H, W, nc, res_t=h.data, x_sl=hsl,
                         y_sl=hsl, **fk(c1))
            if save is not None:                              # keep h_k and u_k: no ping-pong reuse
                save += [h, u]
                h, u, h2 = h2, new(hid), new(hid)
            else:
                h, h2 = h2, h
        if save is not None:
            save.append(h)
        yt = new(cout)
        conv_tangent(h.data, 0, *hd, convf.weight, 1, yt.data, *pn(cout), B, hid, cout, H, W, nc, fmode=F_RELU,
                     f=acts[-1], x_sl=hsl, **fs)
        return yt
    lins = [m for m in net if isinstance(m, nn.Linear)]
    x_t, x_off, x_ci, cin = T.data, view.chan_off * B * nc, view.chan_step * B * nc, view.cin
    fmode, f, f_px = F_NONE, None, 0
    out = None
    for i, lin in enumerate(lins):
        out = Tangent(B, lin.out_features, nc, "fmajor", dev)
        conv_tangent(x_t, x_off, 0, x_ci, nc, lin.weight, 1, out.data, 0, B * nc, nc, 1, cin, lin.out_features, 1, B, nc,
                     fmode=fmode, f=f, f_np=0, f_ci=1, f_px=f_px)
        if i < len(acts):
            fmode, f, f_px = F_TANH, acts[i], lin.out_features
        x_t, x_off, x_ci, cin = out.data, 0, B * nc, lin.out_features
        if save is not None and i + 1 < len(lins):
            save.append(out)
    return out


class GradPool(dict):
    """``grads`` dictionary (parameter -> gradient tensor) whose tensors are slices of ONE zero-filled buffer: a training step
    touches ~470 parameter tensors, and a ``zeros_like`` each was ~470 fill launches of 4 us per backward pass.  Only the parameters a
    backward pass actually reaches get an entry, exactly like the plain dict (parameters outside the graph keep ``grad = None``, as in
    the reference's autograd)."""

    def __init__(self, params, device):
        super().__init__()
        n = sum((p.numel() + 3) // 4 * 4 for p in params)             # 16-byte aligned slices
        self._pool = torch.zeros(max(n, 4), dtype=torch.float32, device=device)
        self._used = 0

    def zeros_for(self, weight):
        n = weight.numel()
        if self._used + n > self._pool.numel():                       # a parameter the pool was not sized for
            return torch.zeros_like(weight, dtype=torch.float32).contiguous()
        g = self._pool[self._used:self._used + n].view(weight.shape)
        self._used += (n + 3) // 4 * 4
        return g


def _grad_of(grads, weight):
    g = grads.get(weight)
    if g is None:
        g = grads[weight] = (grads.zeros_for(weight) if isinstance(grads, GradPool)
                             else torch.zeros_like(weight, dtype=torch.float32).contiguous())
    return g


def net_cotangent(net, YC, view, acts, Ct, saved=None, grads=None, cross=None):
    """Reverse sweep through the coupler network (the adjoint of ``net_tangent``): ``YC`` is the cotangent of the network's
    raw output; the cotangent of its input is ACCUMULATED into the rows of ``Ct`` the network reads (``view``).
    The adjoint of "factor, then conv" is "transposed conv, then factor": transposed / tap-flipped packs
    (``cmf_pack_weight(transpose=1)``) and the OUTPUT-side factor of ``cmf_conv_tangent`` (fp32 MFMA kernel), so every
    stored tensor is the true cotangent of a layer output.
    Training (``saved`` = the list ``net_tangent(save=)`` filled, ``grads`` = dict parameter -> gradient tensor): the weight
    gradient of every layer, ``dW += sum c_out (x) F x_in`` (``cmf_conv_tangent_wgrad``), is accumulated into ``grads``.
    Biases do not act on tangents and get no gradient here.  ``cross`` (MLP networks, a dict): receives, per hidden layer
    index, the cotangent of the primal tanh activation from the tangent rule's second-order term (``cmf_tanh_cross_terms``)."""
    geo, B, nc, dev = view.geom, Ct.B, Ct.nc, Ct.data.device
    f32 = dict(transpose=True, precision="f32")
    train = saved is not None
    assert not train or grads is not None
    if net.kind == "resnet":
        conv0, blocks, convf = _resnet_parts(net)
        hid, cout, H, W, HW = conv0.out_channels, convf.out_channels, geo.H, geo.W, geo.HW
        assert getattr(acts, "f_group", 1) == 1, "the reverse sweep reads per-sample activations"
        new = lambda c: Tangent(B, c * HW, nc, "panel", dev)
        pn = lambda c: (c * HW * nc, HW * nc, nc)          # (np, chan, px) strides of a panel with c channels
        hd, hsl = (hid * HW * nc, 16, hid * nc), hid * 16  # the forward pass's slice-major hidden layout (net_tangent)
        fa = dict(fo_np=hid * HW, fo_co=HW, fo_px=1, fomode=F_RELU)
        fr = dict(fmode=F_RELU, f_np=hid * HW, f_ci=HW, f_px=1)
        # The hidden 3x3 transposed convs run on the split-precision kernel when the forward ones do: no input factor, relu' as
        # an output BIT MASK (derived from the float activations), cotangents slice-major like the forward tangents; the skip
        # connection is a separate accumulate (the kernel's residual input would be masked with the product).
        split = cfg().tangent == "bf16x3" and _shape_ok_bf16x3(9, hid, W, True, H, hid) and hid % 64 == 0
        cd, csl = (hd, hsl) if split else (pn(hid), 16)    # layout of the hidden cotangents
        if train:
            t_in, hs, us = saved[0], saved[1::2], saved[2::2]  # h_0 .. h_K, u_0 .. u_{K-1}
            assert len(hs) == len(blocks) + 1 and len(us) == len(blocks)
            # y = convf(relu'(a_last) . h_K)
            conv_tangent_wgrad(hs[-1].data, 0, *hd, YC.data, 0, *pn(cout), _grad_of(grads, convf.weight), 1, B, hid, cout, H, W, nc,
                               f=acts[-1], x_sl=hsl, **fr)
        # y = convf(relu'(a_last) . h)  ->  c_h = relu'(a_last) . convf^T(c_y)
        ch = new(hid)
        conv_tangent(YC.data, 0, *pn(cout), convf.weight, 1, ch.data, *cd, B, cout, hid, H, W, nc, fo=acts[-1], y_sl=csl, **fa, **f32)
        u, ch2 = new(hid), (None if split else new(hid))      # the in-place skip connection needs no second cotangent buffer
        for k in reversed(range(len(blocks))):
            blk, a_in, c1 = blocks[k], acts[2 * k], acts[2 * k + 1]
            # h2 = h + conv2(relu'(c1) . u), u = conv1(relu'(a_in) . h):
            #   c_u = relu'(c1) . conv2^T(c_h2);   c_h = c_h2 + relu'(a_in) . conv1^T(c_u)
            # split kernels: relu' of both activations as bit masks, shared by the weight gradients (input factor) and the
            # transposed convs (output mask)
            bits_of = lambda t: t if isinstance(t, BitMask) else relu_bits(t)      # ActList: written by the primal pass
            b_c1, b_in = (bits_of(c1), bits_of(a_in)) if split else (None, None)
            assert split or not isinstance(c1, BitMask), "bit-mask activations need the split-precision reverse sweep"
            wf = lambda act, bits: dict(f=bits) if split and nc % 32 == 0 else dict(f=act, **fr)   # (the fp32 kernel reads floats)
            if train:
                conv_tangent_wgrad(us[k].data, 0, *hd, ch.data, 0, *cd, _grad_of(grads, blk.conv2.weight), 9, B, hid, hid, H, W,
                                   nc, x_sl=hsl, y_sl=csl, **wf(c1, b_c1))
            if split:
                conv_tangent(ch.data, 0, *cd, blk.conv2.weight, 9, u.data, *cd, B, hid, hid, H, W, nc, fo=b_c1,
                             transpose=True, x_sl=csl, y_sl=csl)
            else:
                conv_tangent(ch.data, 0, *cd, blk.conv2.weight, 9, u.data, *cd, B, hid, hid, H, W, nc, fo=c1, **fa, **f32)
            if train:
                conv_tangent_wgrad(hs[k].data, 0, *hd, u.data, 0, *cd, _grad_of(grads, blk.conv1.weight), 9, B, hid, hid, H, W,
                                   nc, x_sl=hsl, y_sl=csl, **wf(a_in, b_in))
            if split:
                # the skip connection IN PLACE: c_h += relu'(a_in) . conv1^T(c_u) -- the kernel starts its accumulators from c_h and
                # the lanes the mask switches off do not store (a separate accumulate pass cost 17 ms of a 235 ms step)
                conv_tangent(u.data, 0, *cd, blk.conv1.weight, 9, ch.data, *cd, B, hid, hid, H, W, nc, fo=b_in,
                             transpose=True, x_sl=csl, y_sl=csl, res_t=ch.data)
            else:
                conv_tangent(u.data, 0, *cd, blk.conv1.weight, 9, ch2.data, *cd, B, hid, hid, H, W, nc, res_t=ch.data,
                             fo=a_in, **fa, **f32)
                ch, ch2 = ch2, ch
        # h0 = conv0(mask . v_in)  ->  Ct[view] += mask . conv0^T(c_h0)
        off = view.chan_off * HW * nc
        m = view.mask
        if train:
            conv_tangent_wgrad(t_in.data, 0, *pn(view.cin), ch.data, 0, *cd, _grad_of(grads, conv0.weight), 9, B, view.cin, hid,
                               H, W, nc, fmode=F_RAW if m is not None else F_NONE, f=m, f_np=0, f_ci=HW, f_px=1, y_sl=csl)
        conv_tangent(ch.data, 0, *cd, conv0.weight, 9, Ct.data, Ct.t_b, view.chan_step * HW * nc, nc, B, hid, view.cin, H, W, nc,
                     y_off=off, res_t=Ct.data, res_off=off, fo=m, fo_np=0, fo_co=HW, fo_px=1,
                     fomode=F_RAW if m is not None else F_NONE, x_sl=csl, **f32)
        return
    lins = [mod for mod in net if isinstance(mod, nn.Linear)]
    # x_i = W_i (phi_{i-1} . x_{i-1}), phi_0 = 1, phi_i = tanh'(h_i):  c_{i-1} = phi_{i-1} . W_i^T c_i  (output-side factor);
    # the chain ends unmasked, accumulated into the rows the network read.  dW_i = sum c_i (x) phi_{i-1} x_{i-1}.
    c_t, cin = YC.data, lins[-1].out_features
    for i in reversed(range(len(lins))):
        lin = lins[i]
        if train:
            xin = saved[i]
            fm = dict(fmode=F_TANH, f=acts[i - 1], f_np=0, f_ci=1, f_px=lin.in_features) if i > 0 else {}
            conv_tangent_wgrad(xin.data, 0, 0, B * nc, nc, c_t, 0, 0, B * nc, nc, _grad_of(grads, lin.weight), 1, 1, lin.in_features,
                               lin.out_features, 1, B, nc, **fm)
        if train and i > 0 and cross is not None:
            # training: the unmasked product first, then phi . (in place) together with the second-order term
            #   d h_{i-1} += -2 h_{i-1} sum_col (W_i^T c_i) x_{i-1}      (phi_{i-1} = 1 - h_{i-1}^2 is read by layer i's tangent rule)
            out = Tangent(B, lin.in_features, nc, "fmajor", dev)
            conv_tangent(c_t, 0, 0, B * nc, nc, lin.weight, 1, out.data, 0, B * nc, nc, 1, cin, lin.in_features, 1, B, nc, **f32)
            dh = torch.zeros_like(acts[i - 1])
            _lib.check(_lib.load().cmf_tanh_cross_terms(_p(out.data), out.t_b, out.t_r, _p(saved[i].data), saved[i].t_b, saved[i].t_r,
                                                        _p(acts[i - 1]), _p(dh), lin.in_features, B, nc, _stream()),
                       "cmf_tanh_cross_terms")
            cross[i - 1] = dh
            c_t, cin = out.data, lin.in_features
            continue
        if i == 0:
            off = view.chan_off * B * nc
            conv_tangent(c_t, 0, 0, B * nc, nc, lin.weight, 1, Ct.data, 0, view.chan_step * B * nc, nc, 1, cin, view.cin, 1, B, nc,
                         y_off=off, res_t=Ct.data, res_off=off, **f32)
        else:
            out = Tangent(B, lin.in_features, nc, "fmajor", dev)
            conv_tangent(c_t, 0, 0, B * nc, nc, lin.weight, 1, out.data, 0, B * nc, nc, 1, cin, lin.in_features, 1, B, nc,
                         fo=acts[i - 1], fo_np=0, fo_co=1, fo_px=lin.in_features, fomode=F_TANH, **f32)
            c_t, cin = out.data, lin.in_features


def net_primal_backward(net, z, view, acts, y, g, dy, dg, grads, dz):
    """Primal backward of a ResNet coupler network (training): given the cotangents ``dy`` of its output
    ``y = sw tanh(u) + sb`` and ``dg`` of the tangent multiplier ``g`` (from ``acl_cross_terms``; may be None), accumulates
    the gradients of every weight, bias and of the ScaledTanh parameters into ``grads`` and the cotangent of the rows of
    ``z`` the network read into ``dz``.  ``acts`` = the per-sample float activations ``net_primal(need_acts=True)`` returned.
    Runs on the tangent-conv kernels with 16 samples in the column slots: transposed packs with the per-column output factor
    (``CMF_F_SELF_RELU`` as fomode), ``cmf_conv_tangent_wgrad`` with the input's own relu, ``cmf_channel_sum`` for biases."""
    assert net.kind == "resnet", "MLP couplers: the tanh layers add second-order cross terms that are not built yet"
    geo, B, dev = view.geom, z.shape[0], z.device
    conv0, blocks, convf = _resnet_parts(net)
    hid, cout, H, W, HW = conv0.out_channels, convf.out_channels, geo.H, geo.W, geo.HW
    # hidden weight gradients on the split-precision kernel: it contracts column PAIRS of 16, so two sample groups are presented as
    # the two slices of one 32-column "sample" (slice stride = group stride) -- the batch is padded to a multiple of 32 then
    pair = cfg().tangent == "bf16x3" and hid % 64 == 0
    Bp = (B + 31) // 32 * 32 if pair else (B + 15) // 16 * 16
    G = Bp // 16

    def grp(t):                                            # (B, C, H, W) -> flat grouped (G, C, HW, 16), zero-padded samples
        t = t.reshape(B, -1)
        if Bp != B:
            t = torch.cat([t, torch.zeros(Bp - B, t.shape[1], dtype=t.dtype, device=dev)])
        return primal_regroup(t.contiguous(), True)

    pn = lambda c: (c * HW * 16, HW * 16, 16)
    new = lambda c: torch.empty(G * c * HW * 16, dtype=torch.float32, device=dev)

    # The 2 K hidden weight gradients of the coupler are independent of each other once their operands exist: they are collected and
    # launched TOGETHER (cmf_conv_tangent_wgrad_bf16x3_batched) when one of them alone would leave most of the chip idle -- at a
    # training shard's 2 - 4 sample groups a problem is 28 - 56 image rows for 256 persistent workgroups (13.7 ms in 320 launches of
    # 43 us per C3 step at 64 samples, 12 ms per C5 step at 32)
    gs = hid * HW * 16                                  # group stride = slice stride
    batch_wgrads = pair and (G // 2) * H < 256 and hid == 64
    pending, pending_bias = [], []

    def flush_wgrads():
        for i in range(0, len(pending), _lib.WGRAD_MAX_BATCH):
            part = pending[i:i + _lib.WGRAD_MAX_BATCH]
            conv_tangent_wgrad_batched([p[0] for p in part], [p[1] for p in part], [p[2] for p in part], 2 * gs, HW * 16, 16,
                                       2 * gs, HW * 16, 16, G // 2, hid, hid, H, W, 32, fmode=F_SELF_RELU, x_sl=gs, y_sl=gs)
        pending.clear()
        for i in range(0, len(pending_bias), _lib.WGRAD_MAX_BATCH):       # the hidden convs' bias gradients: one launch too
            part = pending_bias[i:i + _lib.WGRAD_MAX_BATCH]
            channel_sum_batched([p[0] for p in part], *pn(hid), G, hid, HW, 16, [p[1] for p in part])
        pending_bias.clear()

    def hidden_bias_grad(gy_g, bias):
        if batch_wgrads:
            pending_bias.append((gy_g, _grad_of(grads, bias)))
        else:
            channel_sum(gy_g, *pn(hid), G, hid, HW, 16, _grad_of(grads, bias))

    def hidden_wgrad(x_g, gy_g, weight):               # 3x3, hid -> hid, the input's own relu
        if batch_wgrads:
            pending.append((x_g, gy_g, _grad_of(grads, weight)))       # the tensors stay alive until the flush
        elif pair:
            conv_tangent_wgrad(x_g, 0, 2 * gs, HW * 16, 16, gy_g, 0, 2 * gs, HW * 16, 16, _grad_of(grads, weight), 9, G // 2, hid, hid,
                               H, W, 32, fmode=F_SELF_RELU, x_sl=gs, y_sl=gs)
        else:
            conv_tangent_wgrad(x_g, 0, *pn(hid), gy_g, 0, *pn(hid), _grad_of(grads, weight), 9, G, hid, hid, H, W, 16, fmode=F_SELF_RELU)

    tr = dict(transpose=True, precision="f32")
    # hidden 3x3 data-gradient convs: the fp16 split kernel's backward form when the primal pass runs on it (KernelConfig.primal), with
    # its own input-range chain: word 0 = max |da| by a reduction, the others raised by the conv that stores the cotangent
    use16 = cfg().primal == "f16x3" and hid % 64 == 0 and _shape_ok_bf16x3(9, hid, W, True, H, hid)
    rng = torch.zeros(2 * len(blocks) + 1, dtype=torch.float32, device=dev) if use16 else None
    tr16 = lambda i: (dict(transpose=True, precision="f16x3", amax_in=rng[i:i + 1], amax_out=rng[i + 1:i + 2]) if use16 else tr)
    self_fo = lambda t: dict(fo=t, fo_np=hid * HW * 16, fo_co=HW * 16, fo_px=16, fomode=F_SELF_RELU)
    du = stanh_backward(dy, dg, y, g, net.weights, net.bias, _grad_of(grads, net.weights).view(-1), _grad_of(grads, net.bias).view(-1))
    kept = getattr(acts, "grouped", None) if Bp == B else None   # ActList: the forward pass's grouped tensors, no regrouping
    gact = lambda i: kept[i] if kept is not None else grp(acts[i])
    du_g, a_g = grp(du), gact(len(acts) - 1)
    # u = convf(relu(a_K)) + bf
    conv_tangent_wgrad(a_g, 0, *pn(hid), du_g, 0, *pn(cout), _grad_of(grads, convf.weight), 1, G, hid, cout, H, W, 16, fmode=F_SELF_RELU)
    channel_sum(du_g, *pn(cout), G, cout, HW, 16, _grad_of(grads, convf.bias))
    da = new(hid)
    conv_tangent(du_g, 0, *pn(cout), convf.weight, 1, da, *pn(hid), G, cout, hid, H, W, 16, **self_fo(a_g), **tr)
    if use16:
        absmax(da, rng[0:1])
    for j, k in enumerate(reversed(range(len(blocks)))):
        blk = blocks[k]
        a_in, c1 = gact(2 * k), gact(2 * k + 1)
        # a' = a + conv2(relu(c1)) + b2,  c1 = conv1(relu(a)) + b1
        hidden_wgrad(c1, da, blk.conv2.weight)
        hidden_bias_grad(da, blk.conv2.bias)
        dc1 = new(hid)
        conv_tangent(da, 0, *pn(hid), blk.conv2.weight, 9, dc1, *pn(hid), G, hid, hid, H, W, 16, **self_fo(c1), **tr16(2 * j))
        hidden_wgrad(a_in, dc1, blk.conv1.weight)
        hidden_bias_grad(dc1, blk.conv1.bias)
        da2 = new(hid)
        conv_tangent(dc1, 0, *pn(hid), blk.conv1.weight, 9, da2, *pn(hid), G, hid, hid, H, W, 16, res_t=da, **self_fo(a_in), **tr16(2 * j + 1))
        da = da2
    flush_wgrads()
    # a_0 = conv0(mask . z[view])   (no bias)
    cin, m = view.cin, view.mask
    rows = z.reshape(B, geo.C, HW)[:, view.chan_off::view.chan_step][:, :cin]
    x0 = grp(rows)
    fm = dict(fmode=F_RAW, f=m, f_np=0, f_ci=HW, f_px=1) if m is not None else {}
    conv_tangent_wgrad(x0, 0, *pn(cin), da, 0, *pn(hid), _grad_of(grads, conv0.weight), 9, G, cin, hid, H, W, 16, **fm)
    dx0 = new(cin)
    fo = dict(fo=m, fo_np=0, fo_co=HW, fo_px=1, fomode=F_RAW) if m is not None else {}
    conv_tangent(da, 0, *pn(hid), conv0.weight, 9, dx0, *pn(cin), G, hid, cin, H, W, 16, **fo, **tr)
    dz.reshape(B, geo.C, HW)[:, view.chan_off::view.chan_step][:, :cin] += primal_regroup(dx0.view(G, -1), False).view(Bp, cin, HW)[:B]


def mlp_primal_backward(net, z, view, acts, dy, grads, dz, dh_extra=None):
    """Primal backward of an MLP coupler network (2-D / tabular couplers and every low-dimensional prior flow) on the same
    kernels as the ResNet path: 16 samples ride in the 16 column slots of the tangent-conv kernels (``cmf_primal_regroup``), so
    per layer  dW += sum d (x) h  is ``cmf_conv_tangent_wgrad`` with taps = 1,  db  is ``cmf_channel_sum``,  W^T d  is
    ``cmf_conv_tangent`` on the transposed pack, and the tanh stage  d = (dh + extra) (1 - a^2)  is ``cmf_tanh_backward``.
    ``acts`` = the tanh outputs ``net_primal`` returned; ``dh_extra`` = the tangent pass's second-order terms
    (``net_cotangent(cross=)``).  Accumulates weight / bias gradients into ``grads`` and the input cotangent into ``dz``."""
    assert net.kind == "mlp" and view.mask is None
    lins = [m for m in net if isinstance(m, nn.Linear)]
    B, dev = z.shape[0], z.device
    Bp = (B + 15) // 16 * 16
    G = Bp // 16

    def grp(t):                                            # (B, F) -> flat grouped (G, F, 16), zero-padded samples
        t = t.reshape(B, -1)
        if Bp != B:
            t = torch.cat([t, torch.zeros(Bp - B, t.shape[1], dtype=t.dtype, device=dev)])
        return primal_regroup(t.contiguous(), True)

    # a grouped tensor (G, c, 16) as ONE image row of G "pixels" (the call shape of the MLP tangent pass): (np, chan, px) strides
    pn = lambda c: (0, 16, c * 16)
    rows = z.reshape(B, -1)[:, view.chan_off::view.chan_step][:, :view.cin]
    hs = [grp(rows)] + [grp(a) for a in acts]              # input of every linear layer, grouped
    d_g = grp(dy.reshape(B, -1))
    lib = _lib.load()
    for i in reversed(range(len(lins))):
        lin = lins[i]
        cin, cout = lin.in_features, lin.out_features
        conv_tangent_wgrad(hs[i], 0, *pn(cin), d_g, 0, *pn(cout), _grad_of(grads, lin.weight), 1, 1, cin, cout, 1, G, 16)
        channel_sum(d_g, *pn(cout), 1, cout, G, 16, _grad_of(grads, lin.bias))
        dh_g = torch.empty(G * cin * 16, dtype=torch.float32, device=dev)
        conv_tangent(d_g, 0, *pn(cout), lin.weight, 1, dh_g, *pn(cin), 1, cout, cin, 1, G, 16, transpose=True, precision="f32")
        if i > 0:
            extra = grp(dh_extra[i - 1]) if dh_extra is not None and (i - 1) in dh_extra else None
            _lib.check(lib.cmf_tanh_backward(_p(dh_g), _p(hs[i]), _p(extra), dh_g.numel(), _p(dh_g), _stream()), "cmf_tanh_backward")
            d_g = dh_g
        else:
            dh = primal_regroup(dh_g.view(G, -1), False).view(Bp, cin)[:B]
            dz.reshape(B, -1)[:, view.chan_off::view.chan_step][:, :view.cin] += dh
