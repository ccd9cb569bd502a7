"""Coupler networks: parameter containers with the reference's state-dict key schema.

The reference evaluates these with ATen ops and hand-written forward-mode rules
(``cmf/models/components/networks.py``, ``jvp_layers.py``, ``couplers.py``).  Here the modules only
own the parameters (same names, shapes and default initialisation, so checkpoints interchange:
SURVEY.md section 8b); all arithmetic is done by the HIP kernels through ``cmf_amd.engine``.

  MLP    ``get_mlp``     networks.py:206-224 -> keys ``{0,2,4,...}.{weight,bias}``
  ResNet ``get_resnet``  networks.py:116-161 -> ``module.0.weight``, ``module.{i}.conv{1,2}.{weight,bias}``,
                                                 ``module.{n+2}.{weight,bias}``, ``weights``, ``bias``
"""
import torch
import torch.nn as nn

__all__ = ["NN_Sequential_JVP", "ResidualBlock", "ScaledTanh2dModule", "ChunkedSharedCoupler", "get_mlp", "get_resnet"]


def _no_torch_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is evaluated by the HIP engine (cmf_amd.engine), not by torch forward()")


class NN_Sequential_JVP(nn.Sequential):
    """Layer container (reference: networks.py:24-32).  ``kind`` is 'mlp' or 'resnet-body'."""
    forward = _no_torch_forward


class ResidualBlock(nn.Module):
    """relu -> conv3x3 -> relu -> conv3x3 -> + skip, no batch norm (networks.py:35-93 with use_batchnorm=False)."""

    def __init__(self, num_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channels, num_channels, 3, padding=1, bias=True)
        self.conv2 = nn.Conv2d(num_channels, num_channels, 3, padding=1, bias=True)

    forward = _no_torch_forward


class ScaledTanh2dModule(nn.Module):
    """out = weights * tanh(module(x)) + bias, per-channel (networks.py:96-113)."""

    def __init__(self, module, num_channels):
        super().__init__()
        self.module = module
        self.weights = nn.Parameter(torch.ones(num_channels, 1, 1))
        self.bias = nn.Parameter(torch.zeros(num_channels, 1, 1))

    forward = _no_torch_forward


def get_mlp(num_input_channels, hidden_channels, num_output_channels):
    layers, prev = [], num_input_channels
    for h in hidden_channels:
        layers += [nn.Linear(prev, h), nn.Tanh()]
        prev = h
    layers.append(nn.Linear(prev, num_output_channels))
    net = NN_Sequential_JVP(*layers)
    net.kind = "mlp"
    return net


def get_resnet(num_input_channels, hidden_channels, num_output_channels):
    if len(set(hidden_channels)) > 1:
        raise ValueError("resnet couplers use one hidden width (networks.py:133-136 builds equal-width blocks)")
    hid = hidden_channels[0] if hidden_channels else num_output_channels
    layers = [nn.Conv2d(num_input_channels, hid, 3, padding=1, bias=False)]
    layers += [ResidualBlock(hid) for _ in hidden_channels]
    layers += [nn.ReLU(), nn.Conv2d(hid, num_output_channels, 1, bias=True)]
    net = ScaledTanh2dModule(NN_Sequential_JVP(*layers), num_output_channels)
    net.kind = "resnet"
    return net


class ChunkedSharedCoupler(nn.Module):
    """One network emits shift (first half of the channels) and log-scale (second half):
    couplers.py:27-59."""

    def __init__(self, shift_log_scale_net):
        super().__init__()
        self.shift_log_scale_net = shift_log_scale_net

    forward = _no_torch_forward
