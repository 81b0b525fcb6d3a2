"""MI355X-native semantic Sat-NeRF render/training core.

Host-side mirror of the reference's operator surface (framework.* / baseline.* / semantic.*) on top
of libsnerf_hip.so (hand-written HIP for gfx950 behind the C-ABI in include/snerf_hip.h).
There is no CPU or eager-PyTorch fallback: importing ``snerf_amd._lib`` fails loudly when the
library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
__version__ = "0.1.0"
