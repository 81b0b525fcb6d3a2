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
    # 2,826,766 parameters + padding of the packed layout, then one two-plane WF16 pack per weight operand (every
    # matrix and its transpose for dX: 4 bytes per element each)
    assert 3 * 2_800_000 < n < 15_000_000
    d1 = spec.desc(4096, 64, _lib.FLAG_TRAIN | _lib.FLAG_F16X1)
    n1 = L.snerf_packed_floats(C.byref(d1))
    assert 2 * 2_800_000 < n1 < n                                          # one-plane packs: half the bytes behind the fp32 region
    assert L.snerf_grad_floats(C.byref(d1)) == L.snerf_grad_floats(C.byref(d))
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
    # the two arithmetic flags exclude each other (flags = 0 is the default arithmetic, f16x2)
    both = ModelSpec().desc(16, 8, _lib.FLAG_F16X2 | _lib.FLAG_F16X1)
    assert L.snerf_workspace_bytes(C.byref(both)) == 0 and b"arithmetic flag" in L.snerf_last_error()
    one = ModelSpec(fc_units=64, feat_last=32).desc(64, 8, _lib.FLAG_TRAIN | _lib.FLAG_F16X1)
    two = ModelSpec(fc_units=64, feat_last=32).desc(64, 8, _lib.FLAG_TRAIN)
    assert 0 < L.snerf_workspace_bytes(C.byref(one)) < L.snerf_workspace_bytes(C.byref(two))   # 2 bytes per stored activation element
    odd = ModelSpec(fc_units=96, feat_last=48).desc(16, 8, _lib.FLAG_F16X1)                       # one plane: LDS stages of 64 columns
    assert L.snerf_workspace_bytes(C.byref(odd)) == 0 and b"fc_units % 64" in L.snerf_last_error()
    assert L.snerf_workspace_bytes(C.byref(ModelSpec(fc_units=96, feat_last=48).desc(16, 8))) > 0
    # the default arithmetic is the same object for a C caller (flags = 0) and for Python's ModelSpec()
    assert L.snerf_workspace_bytes(C.byref(ModelSpec().desc(64, 8))) == L.snerf_workspace_bytes(C.byref(ModelSpec().desc(64, 8, _lib.FLAG_F16X2)))
    assert L.snerf_version() == 5


def test_plan_builder_over_model_variants_and_null_arguments():
    """The host code behind the C-ABI (plan / table builders, argument checks) over every model variant, pass kind and
    arithmetic; null and misaligned arguments of the hot calls come back as error codes with a message, never as a crash.
    `make -C snerf_amd/csrc asan-test` runs this file against the ASAN + UBSAN host build of the library."""
    from snerf_amd import _lib
    from snerf_amd.ops import ModelSpec
    L = _lib.lib()
    seen = set()
    for kw in ({}, {"siren": False}, {"model": "satnerf"} if "model" in ModelSpec.__dataclass_fields__ else {},
               {"use_separate_beta_for_s": True}, {"use_tj_for_s": True}, {"use_tj_instead_of_beta": True}, {"fc_units": 64}, {"fc_units": 128, "fc_layers": 4, "fc_skips": (2,)}):
        try:
            spec = ModelSpec(**kw)
        except TypeError:
            continue
        for flags in (0, _lib.FLAG_TRAIN, _lib.FLAG_SC_PASS, _lib.FLAG_TRAIN | _lib.FLAG_SC_PASS, _lib.FLAG_TRAIN | _lib.FLAG_F16X1, _lib.FLAG_F16X1, _lib.FLAG_F16X1 | _lib.FLAG_TRAIN | _lib.FLAG_SC_PASS):
            for N, S in ((1, 1), (77, 7), (4096, 64), (2048, 130)):
                d = spec.desc(N, S, flags)
                n, g, w = L.snerf_packed_floats(C.byref(d)), L.snerf_grad_floats(C.byref(d)), L.snerf_workspace_bytes(C.byref(d))
                assert n > 0 and w > 0 and 0 < g <= n, (kw, flags, N, S, L.snerf_last_error())
                seen.add((n, g))
    assert len(seen) >= 4
    d = ModelSpec().desc(64, 8, _lib.FLAG_TRAIN)
    # hot calls with null / misaligned arguments: an error code and a message (no device work is reached)
    assert L.snerf_pack_params(C.byref(d), None, None, None) != 0 and L.snerf_last_error()
    assert L.snerf_forward(C.byref(d), None, None, None, None, 0, None) != 0
    assert L.snerf_backward(C.byref(d), None, None, None, None, None, None, None, 0, None) != 0
    assert L.snerf_unpack_grads(C.byref(d), None, None, 0, None) != 0
    assert L.snerf_grad_floats(None) == 0 and L.snerf_workspace_bytes(None) == 0
    bad = ModelSpec().desc(-5, 64)
    assert L.snerf_grad_floats(C.byref(bad)) == 0 and L.snerf_workspace_bytes(C.byref(bad)) == 0


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


def test_graft_entry_build_runs():
    """The driver's build check (`__graft_entry__.build()`: make for gfx950, import, ABI version): it must not lag behind the
    header (it once asserted the previous ABI version while every other test was green)."""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()
