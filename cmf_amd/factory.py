"""Schema (list of layer dicts) -> nested Density modules.

Same construction order, nesting and wrapping rules as the reference's
``cmf/models/factory.py:55-162`` (so that ``state_dict()`` keys coincide), restricted to the layer
types a non-square model can contain: dequantization, scalar-mult / scalar-add / logit,
non-square-head, acl (checkerboard / split-channel / alternating-channel with shared MLP or ResNet
couplers), squeeze, flatten, split, non-square-base, affine, and the nsf prior's rand-channel-perm / linear / nsf-ar.
"""
import numpy as np
import torch

from .bijections import (AffineBijection, AlternatingChannelwiseAffineCouplingBijection,
                         AutoregressiveRationalQuadraticSplineBijection, Checkerboard2dAffineCouplingBijection,
                         LogitBijection, LULinearBijection, RandomChannelwisePermutationBijection, ScalarAdditionBijection,
                         ScalarMultiplicationBijection, SplitChannelwiseAffineCouplingBijection, Squeeze2dBijection,
                         ViewBijection)
from .densities import (BijectionDensity, DataParallelDensity, DequantizationDensity, DiagonalGaussianDensity,
                        ManifoldFlowHeadDensity, NonSquareHeadDensity, NonSquareTailDensity, SplitDensity)
from .networks import ChunkedSharedCoupler, get_mlp, get_resnet

__all__ = ["get_density", "get_density_recursive"]


def get_density(schema, x_train):
    """``x_train`` is only used for its per-sample shape (the reference's passthrough-before-eval
    wrapper, the one consumer of the data itself, is not part of non-square models)."""
    x_shape = tuple(x_train.shape[1:])
    if schema and schema[0]["type"] == "passthrough-before-eval":
        raise ValueError("passthrough-before-eval is not used by non-square models")
    density = get_density_recursive(schema, x_shape)
    if x_shape[0] != 2:            # factory.py:76-81: wrapped for every dataset whose first dim is not 2
        density = DataParallelDensity(density)
    return density


def _standard_gaussian(x_shape):
    return DiagonalGaussianDensity(mean=torch.zeros(x_shape), stddev=torch.ones(x_shape), num_fixed_samples=64)


def get_density_recursive(schema, x_shape):
    if not schema:
        return _standard_gaussian(x_shape)
    layer, rest = schema[0], schema[1:]
    kind = layer["type"]
    if kind == "dequantization":
        return DequantizationDensity(density=get_density_recursive(rest, x_shape))
    if kind == "split":
        half = (x_shape[0] // 2, *x_shape[1:])
        return SplitDensity(density_1=get_density_recursive(rest, half), density_2=_standard_gaussian(half), dim=1,
                            non_square=layer["non_square"])
    if kind == "non-square-head":
        d = layer["latent_dimension"]
        max_cg = min(layer["max_cg_iterations"], d) if layer["max_cg_iterations"] else d
        cls = ManifoldFlowHeadDensity if layer["m_flow"] else NonSquareHeadDensity
        return cls(prior=get_density_recursive(rest, x_shape), regularization_param=layer["regularization_param"],
                   log_jacobian_method=layer["log_jacobian_method"], x_shape=x_shape,
                   hutchinson_distribution=layer["hutchinson_distribution"],
                   num_hutchinson_samples=layer["hutchinson_samples"], max_cg_iterations=max_cg,
                   cg_tolerance=layer["cg_tolerance"])
    if kind == "non-square-base":
        d = layer["latent_dimension"]
        return NonSquareTailDensity(prior=get_density_recursive(rest, (d,)), x_shape=x_shape, latent_dimension=d,
                                    detach_before_prior=layer["m_flow"])
    bijection = get_bijection(layer, x_shape)
    if layer.get("num_u_channels", 0) != 0:
        raise ValueError("non-square models have num_u_channels == 0 (config.py:46,72-79)")
    return BijectionDensity(bijection=bijection, prior=get_density_recursive(rest, bijection.z_shape))


def get_bijection(layer, x_shape):
    kind = layer["type"]
    if kind == "acl":
        return _get_acl(layer, x_shape)
    if kind == "squeeze":
        return Squeeze2dBijection(x_shape=x_shape, factor=layer["factor"])
    if kind == "flatten":
        return ViewBijection(x_shape=x_shape, z_shape=(int(np.prod(x_shape)),))
    if kind == "logit":
        return LogitBijection(x_shape=x_shape)
    if kind == "scalar-mult":
        return ScalarMultiplicationBijection(x_shape=x_shape, value=layer["value"])
    if kind == "scalar-add":
        return ScalarAdditionBijection(x_shape=x_shape, value=layer["value"])
    if kind == "affine":
        return AffineBijection(x_shape=x_shape, per_channel=layer["per_channel"])
    # the nsf prior of the low-dimensional flow (factory.py:287-314, schemas.py:87-103); parity unpinned (bijections.py)
    if kind == "rand-channel-perm":
        return RandomChannelwisePermutationBijection(x_shape=x_shape)
    if kind == "linear":
        assert len(x_shape) == 1
        return LULinearBijection(num_input_channels=x_shape[0])
    if kind == "nsf-ar":
        assert len(x_shape) == 1 and layer["activation"] == "relu" and layer["dropout_probability"] == 0.
        return AutoregressiveRationalQuadraticSplineBijection(
            num_input_channels=x_shape[0], num_hidden_layers=layer["num_hidden_layers"],
            num_hidden_channels=layer["num_hidden_channels"], num_bins=layer["num_bins"], tail_bound=layer["tail_bound"])
    raise ValueError(f"layer type {kind!r} is outside the non-square hot path")


def _get_coupler(input_shape, num_channels_per_output, config):
    if config["independent_nets"]:
        raise ValueError("non-square configs use shared shift/log-scale nets")
    net = config["shift_log_scale_net"]
    cin, cout = input_shape[0], 2 * num_channels_per_output
    if net["type"] == "mlp":
        assert len(input_shape) == 1 and net["activation"] == "tanh"
        return ChunkedSharedCoupler(get_mlp(cin, net["hidden_channels"], cout))
    if net["type"] == "resnet":
        assert len(input_shape) == 3
        if net.get("batchnorm", True):
            raise ValueError("batch-norm couplers are off for every non-square config (images.py:127-128)")
        return ChunkedSharedCoupler(get_resnet(cin, net["hidden_channels"], cout))
    raise ValueError(f"coupler net {net['type']!r} is outside the hot path")


def _get_acl(config, x_shape):
    C = x_shape[0]
    assert config["num_u_channels"] == 0
    if config["mask_type"] == "checkerboard":
        return Checkerboard2dAffineCouplingBijection(
            x_shape=x_shape, coupler=_get_coupler(x_shape, C, config["coupler"]), reverse_mask=config["reverse_mask"])

    def coupler_factory(num_passthrough_channels):
        return _get_coupler((num_passthrough_channels, *x_shape[1:]), C - num_passthrough_channels, config["coupler"])

    if config["mask_type"] == "alternating-channel":
        return AlternatingChannelwiseAffineCouplingBijection(x_shape, coupler_factory, config["reverse_mask"])
    if config["mask_type"] == "split-channel":
        return SplitChannelwiseAffineCouplingBijection(x_shape, coupler_factory, config["reverse_mask"])
    raise ValueError(f"Invalid mask type {config['mask_type']}")
