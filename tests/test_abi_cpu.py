"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/snerf_hip.h declares, and its host-only entry points behave (no GPU work here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "snerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(snerf_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from snerf_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/snerf_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_struct_sizes_match_header():
    """ctypes mirrors must match the C layout (64-bit pointers, 4-byte ints)."""
    from snerf_amd import _lib
    assert C.sizeof(_lib.SnerfDesc) == 16 * 4
    assert C.sizeof(_lib.SnerfParams) == 8 * (2 * 16 + 12 + 8 + 12)
    assert C.sizeof(_lib.SnerfInputs) == 8 * 9
    assert C.sizeof(_lib.SnerfOutputs) == 8 * 13
    assert C.sizeof(_lib.SnerfOutGrads) == 8 * 11


def test_host_only_sizes_and_errors():
    from snerf_amd import _lib
    from snerf_amd.ops import ModelSpec
    L = _lib.lib()
    spec = ModelSpec()
    d = spec.desc(4096, 64, _lib.FLAG_TRAIN)
    n = L.snerf_packed_floats(C.byref(d))
    # 2,826,766 parameters + padding of the packed layout + K-contiguous transposes for the dX GEMMs
    assert 2 * 2_500_000 < n < 15_000_000  # + three bf16 planes of the fp32 region
    train = L.snerf_workspace_bytes(C.byref(d))
    d.flags = 0
    infer = L.snerf_workspace_bytes(C.byref(d))
    assert 0 < infer < train < 20 * 2**30
    bad = spec.desc(0, 64)
    assert L.snerf_workspace_bytes(C.byref(bad)) == 0
    assert b"n_rays" in L.snerf_last_error()
    bad = ModelSpec(fc_units=520).desc(16, 8)
    assert L.snerf_packed_floats(C.byref(bad)) == 0
    assert b"fc_units" in L.snerf_last_error()
    # arithmetic flags are exclusive (flags = 0 is the default arithmetic, f16x2); split3_bwd2 may name its forward mode
    sel = (_lib.FLAG_F16X2, _lib.FLAG_SPLIT3, _lib.FLAG_FP32_MFMA, _lib.FLAG_BF16, _lib.FLAG_BF16X3, _lib.FLAG_BWD_BF16X3)
    for a in sel:
        for b in sel:
            d = ModelSpec().desc(16, 8, a | b)
            ok = a == b or {a, b} == {_lib.FLAG_SPLIT3, _lib.FLAG_BWD_BF16X3}
            assert (L.snerf_workspace_bytes(C.byref(d)) != 0) == ok, (a, b)
            if not ok:
                assert b"arithmetic flag" in L.snerf_last_error()
    # the default arithmetic is the same object for a C caller (flags = 0) and for Python's ModelSpec()
    assert L.snerf_workspace_bytes(C.byref(ModelSpec().desc(64, 8))) == L.snerf_workspace_bytes(C.byref(ModelSpec().desc(64, 8, _lib.FLAG_F16X2)))
    assert L.snerf_version() == 2


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the HIP path raises instead of computing on the host."""
    import torch
    from snerf_amd import ops
    spec = ops.ModelSpec(fc_units=32, feat_last=16)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.params_struct(spec, {n: torch.zeros(4) for n in spec.param_names()})


def test_param_names_match_reference_state_dict():
    from oracle import snerf_oracle as O
    from snerf_amd.ops import ModelSpec
    for cfg in (O.OracleCfg(), O.OracleCfg(use_separate_beta_for_s=True), O.OracleCfg(model="satnerf")):
        sem = cfg.model == "semantic"
        spec = ModelSpec(n_freq=10 if sem else 0, n_classes=5 if sem else 0,
                         use_separate_beta_for_s=cfg.use_separate_beta_for_s)
        assert spec.param_names() == list(O.param_shapes(cfg).keys())
