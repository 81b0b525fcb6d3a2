"""The SIREN trunk as one persistent launch (csrc/bsp_trunk.hip; one-plane passes of the W = 512 model) against the
launch-per-layer path it replaces (reference: semantic/models/rs_semantic.py:325-334 inside the chunk loop :63-78).

Both paths issue the same MFMA sequence on the same operands per accumulator, the same FMA / v_sin_f32 / fp16 conversion per element and
the same partial sums of sigma's projection, so the comparison is BIT FOR BIT on every rendered tensor -- and the launch-per-layer path is
the one the oracle-based tests of tests/test_gpu_kernels.py / test_gpu_configs.py hold to the reference (one-plane bars: UNPINNED, the
reference publishes no half-precision numbers)."""
import pytest
import torch

from oracle import snerf_oracle as O
from tests.test_gpu_kernels import _dev, _gpu_params, _spec

pytestmark = pytest.mark.gpu


@pytest.fixture
def lib():
    from snerf_amd import _lib
    L = _lib.lib()
    yield L
    L.snerf_test_set_trunk_fusion(1)
    L.snerf_test_set_kc_grid(0)


def _render(cfg, gp, b, dev, sc):
    from snerf_amd import ops
    spec = _spec(cfg)
    rays, extras, u = b["rays"].to(dev), b["extras"].to(dev), b["u"].to(dev)
    t = torch.zeros(rays.shape[0], cfg.t_embedding_tau, device=dev) + 0.25
    zs = torch.linspace(0, 1, cfg.n_samples).to(dev)
    with torch.no_grad():
        return ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, None, sc_pass=sc)


@pytest.mark.parametrize("n_rays,n_samples,grid", [(37, 64, 0), (37, 64, 3), (300, 24, 2), (2048, 64, 0)])
def test_fused_trunk_equals_layer_per_launch_bit_for_bit(n_rays, n_samples, grid, lib, monkeypatch):
    """ragged tiles (37 x 64 = 18.5 tiles of 128 points, 300 x 24 = 56.25), forced persistent grids of 3 / 2 workgroups (every workgroup
    walks many tiles and draws them from the counter), and 1,024 tiles on the default grid; main and solar-correction pass"""
    from snerf_amd import ops, _lib
    dev = _dev()
    monkeypatch.setattr(ops, "BASE_FLAGS", _lib.FLAG_F16X1)
    cfg = O.OracleCfg(n_samples=n_samples)              # W = 512, 8 layers, skip at 4, SIREN
    gp = _gpu_params(O.init_params_numpy(cfg, 21), dev)
    b = O.batch_to_torch(O.synthetic_batch(n_rays, n_samples, seed=n_rays))
    for sc in (False, True):
        lib.snerf_test_set_kc_grid(grid)
        lib.snerf_test_set_trunk_fusion(0)
        ref = _render(cfg, gp, b, dev, sc)
        lib.snerf_test_set_trunk_fusion(1)
        got = _render(cfg, gp, b, dev, sc)
        again = _render(cfg, gp, b, dev, sc)
        assert set(got) == set(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (sc, k, float((got[k].float() - ref[k].float()).abs().max()))
            assert torch.equal(again[k], got[k]), (sc, k)                   # and run to run
        assert torch.isfinite(got["rgb" if not sc else "sun"]).all()


@pytest.mark.parametrize("n_rays,n_samples,grid", [(37, 64, 0), (150, 24, 3), (1024, 64, 0)])
def test_fused_trunk_training_pass_equals_layer_per_launch_bit_for_bit(n_rays, n_samples, grid, lib, monkeypatch):
    """training passes: every layer's planes, exponents and sign words of cos leave the fused launch for the backward pass -- the
    rendered tensors AND every parameter gradient (the backward consumes what the fused forward stored) equal the launch-per-layer
    path's bit for bit, main pass and solar-correction pass"""
    from snerf_amd import ops, _lib
    dev = _dev()
    monkeypatch.setattr(ops, "BASE_FLAGS", _lib.FLAG_F16X1)
    cfg = O.OracleCfg(n_samples=n_samples)
    spec = _spec(cfg)
    pn = O.init_params_numpy(cfg, 22)
    b = O.batch_to_torch(O.synthetic_batch(n_rays, n_samples, seed=n_rays + 1))
    rays, extras, u = b["rays"].to(dev), b["extras"].to(dev), b["u"].to(dev)
    zs = torch.linspace(0, 1, n_samples).to(dev)
    lib.snerf_test_set_kc_grid(grid)

    def run(on, sc):
        lib.snerf_test_set_trunk_fusion(on)
        gp = _gpu_params(pn, dev, requires_grad=True)
        t = (torch.zeros(n_rays, cfg.t_embedding_tau, device=dev) + 0.25).requires_grad_(True)
        res = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, None, sc_pass=sc)
        g = torch.Generator().manual_seed(5)
        loss = 0.0
        for k in sorted(res):
            if k not in ("z_vals", "semantic_label") and res[k].requires_grad:
                loss = loss + (res[k] * torch.rand(res[k].shape, generator=g).to(dev)).sum()
        loss.backward()
        return {k: v.detach() for k, v in res.items()}, {k: v.grad for k, v in gp.items()}, t.grad

    for sc in (False, True):
        r0, g0, t0 = run(0, sc)
        r1, g1, t1 = run(1, sc)
        for k in r0:
            assert torch.equal(r1[k], r0[k]), (sc, k)
        n = 0
        for k in g0:
            if g0[k] is None:
                assert g1[k] is None, k
                continue
            assert torch.equal(g1[k], g0[k]), (sc, k, float((g1[k] - g0[k]).abs().max()))
            n += 1
        assert n >= 20
        assert (t0 is None and t1 is None) or torch.equal(t0, t1)


def test_fused_trunk_is_taken_only_where_it_applies(lib, monkeypatch):
    """two planes, other widths: the plan keeps the launch-per-layer path (and the results do not depend on the switch)"""
    from snerf_amd import ops, _lib
    dev = _dev()
    b = O.batch_to_torch(O.synthetic_batch(40, 16, seed=2))
    for flags, cfg in ((0, O.OracleCfg(n_samples=16)), (_lib.FLAG_F16X1, O.OracleCfg(n_samples=16, fc_units=256))):
        monkeypatch.setattr(ops, "BASE_FLAGS", flags)
        gp = _gpu_params(O.init_params_numpy(cfg, 4), dev)
        lib.snerf_test_set_trunk_fusion(0)
        ref = _render(cfg, gp, b, dev, False)
        lib.snerf_test_set_trunk_fusion(1)
        got = _render(cfg, gp, b, dev, False)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
