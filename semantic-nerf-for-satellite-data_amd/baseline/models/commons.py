"""SIREN pieces -- mirror of baseline/models/commons.py:5-74.

The activation and the positional mapping are computed inside the fused HIP kernels (GEMM epilogue /
encode kernel); the classes here only carry the hyper-parameters and the initialisers so that
nn.Module trees, state_dict keys and init distributions equal the reference's."""
import numpy as np
import torch


def sine_init(m):
    with torch.no_grad():
        if hasattr(m, "weight"):
            n = m.weight.size(-1)
            m.weight.uniform_(-np.sqrt(6 / n), np.sqrt(6 / n))


def first_layer_sine_init(m):
    with torch.no_grad():
        if hasattr(m, "weight"):
            n = m.weight.size(-1)
            m.weight.uniform_(-1 / n, 1 / n)


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
