"""ctypes binding of libcmf_amd.so (C ABI declared in include/cmf_amd.h).

The product path has no CPU fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcmf_amd.so")

F_NONE, F_RELU, F_TANH, F_RAW, F_SELF_RELU, F_RELU_BITS = 0, 1, 2, 3, 4, 5
O_NONE, O_TANH, O_STANH = 0, 1, 2

_fp = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
_ll = C.c_longlong
_i = C.c_int
_f = C.c_float
_d = C.c_double


class ConvTangentArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_np", _ll), ("x_ci", _ll), ("x_px", _ll),
                ("f", _fp), ("f_np", _ll), ("f_ci", _ll), ("f_px", _ll), ("fmode", _i),
                ("w", _fp),
                ("y", _fp), ("y_np", _ll), ("y_co", _ll), ("y_px", _ll),
                ("r", _fp), ("r_np", _ll), ("r_co", _ll), ("r_px", _ll),
                ("np", _i), ("cin", _i), ("cout", _i), ("H", _i), ("W", _i), ("nc", _i), ("taps", _i),
                ("bias", _fp), ("f_group", _i), ("x_sl", _ll), ("y_sl", _ll), ("r_sl", _ll),
                ("fo", _fp), ("fo_np", _ll), ("fo_co", _ll), ("fo_px", _ll), ("fomode", _i),
                ("mask_out", _fp), ("mask_np", _ll), ("amax_in", _fp), ("amax_out", _fp), ("live", _i)]


class ConvPrimalArgs(C.Structure):
    _fields_ = [("x", _fp), ("x_b", _ll), ("x_c", _ll), ("x_px", _ll),
                ("f", _fp), ("f_c", _ll), ("f_px", _ll), ("imode", _i),
                ("w", _fp), ("bias", _fp), ("sw", _fp), ("sb", _fp),
                ("y", _fp), ("y_b", _ll), ("y_c", _ll), ("y_px", _ll),
                ("g", _fp),
                ("r", _fp), ("r_b", _ll), ("r_c", _ll), ("r_px", _ll),
                ("omode", _i),
                ("B", _i), ("cin", _i), ("cout", _i), ("H", _i), ("W", _i), ("taps", _i)]


MLP_MAX_LAYERS = 8
WGRAD_MAX_BATCH = 16


class MlpCouplerArgs(C.Structure):
    _fields_ = [("z", _fp), ("z_b", _ll), ("t", _fp), ("t_f", _ll), ("w", _fp),
                ("zi", _fp), ("si", _fp), ("ti", _fp), ("n_mod", _i),
                ("B", _i), ("cin", _i), ("chan_off", _i), ("chan_step", _i), ("n_layers", _i),
                ("width", _i * (MLP_MAX_LAYERS + 1)), ("w_off", _ll * MLP_MAX_LAYERS), ("decode", _i), ("lj", _fp), ("ncols", _i)]


#: every symbol include/cmf_amd.h declares -> (restype, argtypes)
SIGNATURES = {
    "cmf_version": (C.c_char_p, []),
    "cmf_pack_weight": (_i, [_fp, _fp, _i, _i, _i, _i, C.POINTER(_ll), _fp]),
    "cmf_pack_weights_batched": (_i, [_fp, _i, _fp]),
    "cmf_conv_tangent": (_i, [C.POINTER(ConvTangentArgs), _fp]),
    "cmf_pack_weight_bf16x3": (_i, [_fp, _fp, _i, _i, C.POINTER(_ll), _fp]),
    "cmf_pack_weight_bf16x3_t": (_i, [_fp, _fp, _i, _i, _i, C.POINTER(_ll), _fp]),
    "cmf_conv_tangent_bf16x3": (_i, [C.POINTER(ConvTangentArgs), _fp]),
    "cmf_pack_weight_f16x3": (_i, [_fp, _fp, _i, _i, _i, C.POINTER(_ll), _fp]),
    "cmf_conv_tangent_f16x3": (_i, [C.POINTER(ConvTangentArgs), _fp]),
    "cmf_conv_tangent_f16x3_item": (_i, [C.POINTER(ConvTangentArgs), _i, _fp]),
    "cmf_absmax": (_i, [_fp, _ll, _fp, _fp]),
    "cmf_conv_tangent_wgrad_ws": (_ll, [C.POINTER(ConvTangentArgs)]),
    "cmf_conv_tangent_wgrad": (_i, [C.POINTER(ConvTangentArgs), _fp, _fp, _fp, _ll, _fp]),
    "cmf_conv_tangent_wgrad_bf16x3": (_i, [C.POINTER(ConvTangentArgs), _fp, _fp, _fp, _ll, _fp]),
    "cmf_conv_tangent_wgrad_bf16x3_batched": (_i, [C.POINTER(ConvTangentArgs), _i, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp), _fp, _ll, _fp]),
    "cmf_primal_regroup": (_i, [_fp, _fp, _i, _ll, _i, _fp]),
    "cmf_conv_primal": (_i, [C.POINTER(ConvPrimalArgs), _fp]),
    "cmf_acl_primal": (_i, [_fp, _ll, _fp, _ll, _fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "cmf_acl_tangent": (_i, [_fp, _ll, _ll, _fp, _ll, _ll, _i, _fp, _ll, _fp, _ll, _fp, _fp, _fp, _fp, _i, _i, _fp]),
    "cmf_acl_cotangent": (_i, [_fp, _ll, _ll, _fp, _ll, _ll, _i, _fp, _ll, _fp, _ll, _fp, _fp, _fp, _fp, _i, _i, _fp]),
    "cmf_acl_primal_backward": (_i, [_fp, _ll, _fp, _ll, _fp, _ll, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "cmf_acl_cross_terms": (_i, [_fp, _ll, _ll, _fp, _ll, _ll, _fp, _ll, _ll, _i, _fp, _ll, _fp, _ll, _fp, _fp, _fp, _fp, _i, _i,
                            _fp, _fp, _fp, _fp]),
    "cmf_gather_primal": (_i, [_fp, _ll, _fp, _ll, _fp, _i, _i, _fp]),
    "cmf_gather_tangent": (_i, [_fp, _ll, _ll, _fp, _ll, _ll, _fp, _i, _i, _i, _fp]),
    "cmf_seed_tangent": (_i, [_fp, _ll, _ll, _fp, _i, _i, _fp, _i, _i, _i, _fp]),
    "cmf_gram_cholesky": (_i, [_fp, _ll, _ll, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "cmf_cholesky_retry": (_i, [_fp, _i, _i, _i, _f, _fp, _fp, _fp, _fp, _fp]),
    "cmf_stanh_backward": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    "cmf_tanh_cross_terms": (_i, [_fp, _ll, _ll, _fp, _ll, _ll, _fp, _fp, _i, _i, _i, _fp]),
    "cmf_tanh_backward": (_i, [_fp, _fp, _fp, _ll, _fp, _fp]),
    "cmf_affine_prior_backward": (_i, [_fp, _ll, _fp, _ll, _fp, _i, _i, _fp, _fp, _fp, _fp]),
    "cmf_accumulate": (_i, [_fp, _fp, _ll, _fp]),
    "cmf_relu_bits": (_i, [_fp, _fp, _i, _i, _i, _fp]),
    "cmf_channel_sum": (_i, [_fp, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _fp, _fp]),
    "cmf_channel_sum_batched": (_i, [C.POINTER(_fp), C.POINTER(_fp), _i, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _fp]),
    "cmf_grad_sqnorm": (_i, [_fp, _ll, _fp, _fp, _fp]),
    "cmf_optimizer_step": (_i, [_i, _fp, _fp, _fp, _fp, _ll, _d, _d, _d, _d, _d, _i, _fp, _f, _fp]),
    "cmf_gram_backward_matrix": (_i, [_fp, _ll, _ll, _i, _i, _i, _i, _fp, _fp, _ll, _ll, _fp]),
    "cmf_gram_backward": (_i, [_fp, _ll, _ll, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _ll, _ll, _fp]),
    "cmf_prehead": (_i, [_fp, _fp, _fp, _fp, _f, _f, _i, _i, _i, _fp]),
    "cmf_prehead_inverse": (_i, [_fp, _fp, _f, _f, _i, _ll, _fp]),
    "cmf_expand_columns": (_i, [_fp, _i, _fp, _i, _fp, _ll, _fp]),
    "cmf_gaussian_logprob": (_i, [_fp, _ll, _i, _i, _fp, _fp]),
    "cmf_affine_prior": (_i, [_fp, _ll, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "cmf_recon_sqerr": (_i, [_fp, _fp, _i, _i, _fp, _fp]),
    "cmf_elbo_combine": (_i, [_fp, _fp, _fp, _fp, _fp, _f, _f, _f, _i, _fp, _fp]),
    "cmf_hutch_cg": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _f, _fp, _fp, _fp, _fp, _fp]),
    "cmf_mlp_coupler": (_i, [C.POINTER(MlpCouplerArgs), _fp]),
    "cmf_pack_mlp_layer": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _fp, C.POINTER(_ll), _fp]),
    "cmf_mlp_hidden_tiles": (_i, [_i]),
    "cmf_rq_spline": (_i, [_fp, _ll, _fp, _i, _i, _i, _f, _i, _i, _fp, _ll, _fp, _fp]),
    "cmf_rq_spline_backward": (_i, [_fp, _ll, _fp, _i, _i, _i, _f, _i, _fp, _ll, _fp, _fp, _ll, _fp, _fp]),
    "cmf_lu_backward": (_i, [_fp, _fp, _fp, _fp, _i, _f, _fp, _i, _fp, _fp, _fp, _fp, _fp]),
    "cmf_gaussian_backward": (_i, [_fp, _fp, _i, _i, _fp, _fp]),
    "cmf_lu_weights": (_i, [_fp, _fp, _fp, _i, _f, _fp, _fp, _fp]),
    "cmf_made_mask_weight": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _fp]),
    "cmf_hutch_metric": (_i, [_fp, _i, _i, _i, _fp, _fp, _fp]),
    "cmf_hutch_cotangent": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    "cmf_hutch_lowrank_cotangent": (_i, [_fp, _i, _i, _i, _fp, _fp, _i, _fp, _fp]),
}

_lib = None
_tls = threading.local()


class _Traced:
    """``load()``'s return value while a ``trace()`` is active on the calling thread: every foreign call is bracketed by two HIP
    events on the current stream and recorded as (symbol, phase label, event, event).  bench.py's per-stage breakdown."""

    def __init__(self, lib, sink):
        self._lib, self._sink = lib, sink

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("cmf_"):
            return fn
        import torch

        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            role = getattr(_tls, "role", None)
            self._sink.append((name + ":" + role if role else name, getattr(_tls, "phase", None), e0, e1))
            return rc
        return call


class trace:
    """``with trace() as records:`` -- every libcmf_amd call of THIS thread inside the block is timed with HIP events (a
    diagnostic mode: two event records per launch cost a few microseconds each); ``phase(label)`` tags the calls."""

    def __enter__(self):
        self.prev = getattr(_tls, "sink", None)
        _tls.sink = []
        return _tls.sink

    def __exit__(self, *exc):
        _tls.sink = self.prev
        return False


class phase:
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        self.prev = getattr(_tls, "phase", None)
        if self.prev is None:                               # the outermost label wins (encode_train calls encode's helpers)
            _tls.phase = self.label

    def __exit__(self, *exc):
        _tls.phase = self.prev
        return False


def tracing():
    """True while a ``trace()`` is active on the calling thread."""
    return getattr(_tls, "sink", None) is not None


class role:
    """``with role("primal"):`` -- calls traced inside are recorded as "<symbol>:primal" (one kernel serving two stages: the fp32
    tangent conv also runs the sample-grouped primal convs of small shards)."""

    def __init__(self, label):
        self.label = label

    def __enter__(self):
        self.prev = getattr(_tls, "role", None)
        _tls.role = self.label

    def __exit__(self, *exc):
        _tls.role = self.prev
        return False


def load():
    """Load libcmf_amd.so and bind every declared symbol; raises if the library or a symbol is missing."""
    global _lib
    sink = getattr(_tls, "sink", None)
    if _lib is not None:
        return _lib if sink is None else _Traced(_lib, sink)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP kernels are the only implementation of this path "
            "(no CPU fallback). Build them with `python -m cmf_amd.build`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "size out of range"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"libcmf_amd: {what} failed: {kind}")
