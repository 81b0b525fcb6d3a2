"""SIREN pieces -- mirror of baseline/models/commons.py:5-74.

The activation and the positional mapping are computed inside the fused HIP kernels (GEMM epilogue /
encode kernel); the classes here only carry the hyper-parameters and the initialisers so that
nn.Module trees, state_dict keys and init distributions equal the reference's."""
import math

import torch


def _uniform_by_fan_in(module, bound_of_fan_in):
    """U(-b, b) on `module.weight` with b a function of the layer's fan-in; biases (and modules without a weight, as
    `Sequential.apply` also visits the activations) keep what they have -- PyTorch's default Linear init."""
    w = getattr(module, "weight", None)
    if w is not None:
        b = float(bound_of_fan_in(w.shape[-1]))
        torch.nn.init.uniform_(w, -b, b)


def sine_init(m):
    """SIREN hidden layers: b = sqrt(6 / fan_in) (baseline/models/commons.py:5-10)"""
    _uniform_by_fan_in(m, lambda n: math.sqrt(6.0 / n))


def first_layer_sine_init(m):
    """SIREN first layer: b = 1 / fan_in (baseline/models/commons.py:13-18)"""
    _uniform_by_fan_in(m, lambda n: 1.0 / n)


class _FusedOnly(torch.nn.Module):
    def forward(self, *a, **k):
        raise NotImplementedError(
            f"{type(self).__name__} is evaluated inside libsnerf_hip.so (snerf_forward); call inference() / "
            "the renderer instead of the per-point module")


class Siren(_FusedOnly):
    def __init__(self, w0=1.0):
        super().__init__()
        self.w0 = w0


class Mapping(_FusedOnly):
    def __init__(self, mapping_size, in_size, logscale=True):
        super().__init__()
        self.N_freqs = mapping_size
        self.in_channels = in_size
        self.out_channels = in_size * (2 * mapping_size + 1)
        if not logscale:
            raise NotImplementedError("only log-scale frequency bands (2^k) are implemented in the HIP encoder")
        self.freq_bands = 2 ** torch.linspace(0, mapping_size - 1, mapping_size)


def get_nl(activation_function):
    return {"relu": torch.nn.ReLU, "siren": Siren}[activation_function]
