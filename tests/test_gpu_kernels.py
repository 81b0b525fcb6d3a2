"""GPU parity tests at the C-ABI level (run with -m gpu on an MI355X).

Every check compares libsnerf_hip.so (through snerf_amd.ops -> ctypes -> C-ABI) with the CPU oracle
(oracle/snerf_oracle.py, itself pinned to the reference by tests/test_oracle_golden.py) on the same
seeded inputs.  Tolerances: 1e-4 absolute on rendered outputs (BASELINE.json north_star), exact class
argmax wherever the oracle's top-2 logit margin exceeds the output tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import snerf_oracle as O
from tests.helpers import load_fixture, fixture_params, fixture_batch, max_abs, rel_err

pytestmark = pytest.mark.gpu

OUT_TOL = 1e-4
# relative L2 per parameter tensor, default (fp32-class) arithmetic: 10x the largest error MEASURED on any tensor of any case
# (1e-6 ... 4e-5; fp32 chains of different summation order on both sides).  A backward kernel that lost 4 bits fails this.
GRAD_REL_TOL = 2e-4
GRAD_ABS_ESCAPE = 1e-5   # x max|reference|: tensors whose gradient is rounding noise around zero (max_abs <= 1e-7 + this * scale)
LABEL_AMBIGUOUS_DISAGREE_MAX = 0.01   # of all rays: labels that differ where the top-2 margin is inside the logits' own fp32 noise


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


# ----------------------------------------------------------------------------------------------------
# one rendering pass vs the oracle
# ----------------------------------------------------------------------------------------------------
def _gpu_params(params_np, dev, requires_grad=False):
    return {k: torch.from_numpy(v).to(dev).requires_grad_(requires_grad) for k, v in params_np.items()}


def _spec(cfg):
    from snerf_amd.ops import ModelSpec
    sem = cfg.model == "semantic"
    return ModelSpec(fc_units=cfg.fc_units, fc_layers=cfg.fc_layers, feat_last=cfg.feat_last,
                     fc_skips=tuple(cfg.fc_skips), n_freq=cfg.mapping_pos_n_freq if sem else 0, siren=cfg.siren,
                     t_dim=cfg.t_embedding_tau, n_classes=cfg.n_classes if sem else 0,
                     sem_sigmoid=sem and cfg.semantic_activation_function == "sigmoid",
                     use_tj_instead_of_beta=sem and cfg.use_tj_instead_of_beta, use_tj_for_s=sem and cfg.use_tj_for_s,
                     use_separate_beta_for_s=sem and cfg.use_separate_beta_for_s,
                     use_separate_tj_for_semantic=sem and cfg.use_separate_tj_for_semantic)


def _hip_render(cfg, gp, emb_g, b, dev, emb_s_g=None):
    """main (+ sc) pass through the C-ABI, assembled like RSSemanticRendering does."""
    from snerf_amd import ops
    spec = _spec(cfg)
    rays, extras, u = b["rays"].to(dev), b["extras"].to(dev), b["u"].to(dev)
    ts = extras[:, 3].long()
    t = emb_g[ts]
    t_s = emb_s_g[ts] if emb_s_g is not None else None
    zs = torch.linspace(0, 1, cfg.n_samples).to(dev)  # host linspace, as the CPU reference computes it
    packed = ops.pack_params(spec, gp)
    res = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, t_s,
                          packed=packed)
    if cfg.sc_lambda > 0:
        sc = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_vals=res["z_vals"]), t, t_s,
                             sc_pass=True, packed=packed)
        res["weights_sc"], res["transparency_sc"], res["sun_sc"] = sc["weights"], sc["transparency"], sc["sun"]
    zv = res.pop("z_vals")
    out = {f"{k}_coarse": v for k, v in res.items()}
    out["_z_vals"] = zv
    return out


def test_given_depths_are_returned_as_the_callers_storage():
    """Contract (ops.render_pass): when the depths are GIVEN (seam 3, the solar-correction pass), results['z_vals'] is the caller's own
    tensor, not a copy -- no copy-out launch.  Main pass result, sc pass input and sc pass result are then ONE buffer: whoever edits it
    in place edits all three (and the depths a later backward reads).  No caller in the package does; this test makes the aliasing explicit."""
    from snerf_amd import ops
    dev = _dev()
    cfg = O.OracleCfg(fc_units=32, n_samples=16)
    spec = _spec(cfg)
    gp = _gpu_params(O.init_params_numpy(cfg, 3), dev)
    b = O.batch_to_torch(O.synthetic_batch(24, 16, seed=9))
    rays, extras, u = b["rays"].to(dev), b["extras"].to(dev), b["u"].to(dev)
    t = torch.zeros(24, cfg.t_embedding_tau, device=dev)
    with torch.no_grad():
        res = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=torch.linspace(0, 1, 16).to(dev), u=u), t, None)
        z_main = res["z_vals"]
        sc = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_vals=z_main), t, None, sc_pass=True)
    assert z_main.data_ptr() != u.data_ptr()                      # sampled depths: a fresh tensor
    assert sc["z_vals"].data_ptr() == z_main.data_ptr()          # given depths: the caller's storage
    assert torch.equal(sc["z_vals"], z_main)


LABEL_STATS = []   # one record per comparison: how many rays qualified for the exact-label check, disagreements among the rest


def _compare_outputs(hip, ora, cfg):
    for k, v in ora.items():
        if k == "semantic_label_coarse":
            # "class argmax bit-exact": the labels must agree wherever the oracle's top-2 margin exceeds twice the MEASURED error
            # of the logits (a smaller margin is a tie inside fp32 noise of either implementation); nearly every ray qualifies,
            # and the disagreements among the few that do not are counted and reported, not hidden
            logits = ora["semantic_logits_coarse"].detach()
            lerr = max_abs(hip["semantic_logits_coarse"].detach().cpu(), logits)
            top2 = logits.topk(2, dim=-1).values
            sure = (top2[:, 0] - top2[:, 1]) > 2 * max(lerr, 1e-6)
            same = hip[k].cpu() == v
            assert bool(same[sure].all()), "class argmax differs on rays with a clear margin"
            frac = float(sure.float().mean())
            assert frac >= 0.99 or sure.numel() < 200, ("margin-qualified fraction", frac)
            n_dis = int((~same[~sure]).sum())
            LABEL_STATS.append({"rays": int(sure.numel()), "qualified": frac, "logit_err": lerr, "ambiguous": int((~sure).sum()),
                                "ambiguous_disagree": n_dis})
            # ties inside fp32 noise cannot be matched bit for bit, but they are bounded: at most 1 % of the rays (one ray on tiny batches)
            assert n_dis <= max(1, int(LABEL_AMBIGUOUS_DISAGREE_MAX * sure.numel())), ("ambiguous labels that disagree", n_dis, int(sure.numel()))
            continue
        err = max_abs(hip[k].detach().cpu(), v.detach())
        assert err <= OUT_TOL, (k, err)


FIXTURES = ["sem_siren_small", "sem_relu_small", "sem_variants_small", "sem_tj_small", "sem_cartreg_small",
            "satnerf_small", "satnerf_relu_small"]


@pytest.mark.parametrize("name", FIXTURES)
def test_forward_matches_oracle_and_golden(name):
    dev = _dev()
    z, meta, cfg = load_fixture(name)
    pn = fixture_params(z, meta, cfg)
    b = fixture_batch(z)
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"]))
    emb_s = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"] + 1)) if cfg.use_separate_tj_for_semantic else None
    with torch.no_grad():
        hip = _hip_render(cfg, _gpu_params(pn, dev), emb.to(dev), b, dev, emb_s.to(dev) if emb_s is not None else None)
        ora = O.render_rays(O.to_torch(pn), emb, cfg, b["rays"], b["extras"], b["u"], emb_s)
    # z_vals are reproduced bit for bit (no FMA contraction in the sampler)
    assert torch.equal(hip["_z_vals"].cpu(), ora["_z_vals"]), max_abs(hip["_z_vals"].cpu(), ora["_z_vals"])
    ora.pop("_z_vals"); hip.pop("_z_vals")
    _compare_outputs(hip, ora, cfg)
    # and directly against the reference's own outputs stored in the fixture
    for k in z.files:
        if k.startswith("out_") and not k.startswith("out_depth_") and k != "out_semantic_label_coarse":
            assert max_abs(hip[k[4:]].cpu(), z[k]) <= OUT_TOL, k


@pytest.mark.parametrize("name", FIXTURES)
def test_backward_matches_oracle_and_golden(name):
    dev = _dev()
    z, meta, cfg = load_fixture(name)
    pn = fixture_params(z, meta, cfg)
    b = fixture_batch(z)
    emb_np = O.init_embedding_numpy(cfg, meta["seed"])
    gp = _gpu_params(pn, dev, requires_grad=True)
    emb_g = torch.from_numpy(emb_np).to(dev).requires_grad_(True)
    emb_s_g = None
    if cfg.use_separate_tj_for_semantic:
        emb_s_g = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"] + 1)).to(dev).requires_grad_(True)
    hip = _hip_render(cfg, gp, emb_g, b, dev, emb_s_g)
    hip.pop("_z_vals")
    bg = {k: v.to(dev) for k, v in b.items()}
    depth_res = None
    if meta["with_depth"]:
        bd = fixture_batch(z, "in_depth_")
        depth_res = _hip_render(cfg, gp, emb_g, bd, dev, emb_s_g)
    # losses evaluated with the oracle's loss restatement on the HIP outputs (loss kernels are tested separately)
    ld = O.training_losses(hip, bg, cfg, meta["epoch"], depth_res)
    for k in ld:
        ref = float(z["loss_" + k])
        assert abs(float(ld[k].detach()) - ref) <= 2e-4 * max(1.0, abs(ref)), (k, float(ld[k].detach()), ref)
    O.total_loss(ld).backward()
    grads = {k: v.grad for k, v in gp.items()}
    grads["model_t.weight"] = emb_g.grad
    if emb_s_g is not None:
        grads["model_t_s.weight"] = emb_s_g.grad
    n = 0
    for k in z.files:
        if not k.startswith("grad_"):
            continue
        g = grads[k[5:]]
        ref = z[k]
        assert g is not None, k
        err = rel_err(g.cpu(), ref)
        scale = float(np.abs(ref).max())
        assert err <= GRAD_REL_TOL or max_abs(g.cpu(), ref) <= 1e-7 + GRAD_ABS_ESCAPE * scale, (k, err, scale)
        n += 1
    assert n >= 20


@pytest.fixture
def kc_grid_3():
    """persistent grid of 3 workgroups for every K-contiguous launch (C-ABI test hook): at 1,024 points x 512 columns = 16 tiles every
    workgroup then walks several tiles and draws them from the pass's tile counters -- which the pass's FIRST kernel clears on the side
    (encode forward, composite backward); the default grid gives every tile a workgroup of its own at this size"""
    from snerf_amd import _lib
    L = _lib.lib()
    L.snerf_test_set_kc_grid(3)
    yield 3
    L.snerf_test_set_kc_grid(0)


def test_full_width_pass_on_a_forced_small_grid(kc_grid_3):
    """the in-pass clearing of the tile counters, pinned at a small size -- the whole forward + backward of the W = 512
    reference fixture with three persistent workgroups per launch (the counters are LIVE: 16 tiles, 3 workgroups), against the same
    reference outputs, losses and gradients as the default grid."""
    test_full_width_forward_backward("sem_siren_full")


@pytest.mark.parametrize("name", ["sem_siren_full", "satnerf_full_c1"])
def test_full_width_forward_backward(name):
    """W=512 (the headline architecture; S=64 semantic, and BASELINE configs[0]'s baseline SatNeRF at S=32) on the 16-ray
    golden cases written by the reference itself: per-ray outputs, losses, grad norms."""
    dev = _dev()
    z, meta, cfg = load_fixture(name)
    pn = fixture_params(z, meta, cfg)
    b = fixture_batch(z)
    gp = _gpu_params(pn, dev, requires_grad=True)
    emb_g = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"])).to(dev).requires_grad_(True)
    hip = _hip_render(cfg, gp, emb_g, b, dev)
    hip.pop("_z_vals")
    for k in z.files:
        if k.startswith("out_") and k != "out_semantic_label_coarse":
            assert max_abs(hip[k[4:]].detach().cpu(), z[k]) <= OUT_TOL, (k, max_abs(hip[k[4:]].detach().cpu(), z[k]))
    # the inference-mode pass (no stored activations, no sign words: other instantiations of the same kernels, incl. the folded
    # sigma / sun projections of full-width SIREN passes) gives the training-mode results bit for bit
    with torch.no_grad():
        inf = _hip_render(cfg, gp, emb_g, b, dev)
    inf.pop("_z_vals")
    for k, v in hip.items():
        assert torch.equal(inf[k], v.detach()), k
    bg = {k: v.to(dev) for k, v in b.items()}
    ld = O.training_losses(hip, bg, cfg, meta["epoch"])
    for k in ld:
        ref = float(z["loss_" + k])
        assert abs(float(ld[k].detach()) - ref) <= 2e-4 * max(1.0, abs(ref)), (k, float(ld[k].detach()), ref)
    O.total_loss(ld).backward()
    grads = {k: v.grad for k, v in gp.items()}
    grads["model_t.weight"] = emb_g.grad
    for k in z.files:
        if k.startswith("gradnorm_"):
            g = grads[k[9:]]
            nrm = float(g.double().norm())
            ref = float(z[k])
            assert abs(nrm - ref) <= GRAD_REL_TOL * max(ref, 1e-9), (k, nrm, ref)
            s = g.detach().cpu().reshape(-1)[:: max(1, g.numel() // 64)][:64]
            rs = z["gradsample_" + k[9:]]
            assert rel_err(s, rs) <= 10 * GRAD_REL_TOL or max_abs(s, rs) <= GRAD_ABS_ESCAPE * float(np.abs(rs).max() + 1e-12), k
    # the reference's FULL gradients of the first and the skip trunk layer, the sun head's first layer, the semantic head's last
    # layer and the embedding (tools/gen_golden.py: FULL_GRADS_SEM), element for element
    n_full = 0
    for k in z.files:
        if k.startswith("grad_"):
            err = rel_err(grads[k[5:]].detach().cpu(), z[k])
            assert err <= GRAD_REL_TOL, (k, err)
            n_full += 1
    assert n_full == (5 if name == "sem_siren_full" else 0)


def test_inference_seam_explicit_xyz():
    dev = _dev()
    from snerf_amd import ops
    for name in ["inference_sem_small", "inference_satnerf_small"]:
        z, meta, cfg = load_fixture(name)
        pn = fixture_params(z, meta, cfg)
        gp = _gpu_params(pn, dev)
        with torch.no_grad():
            r = ops.render_pass(_spec(cfg), gp, ops.PassInputs(
                sun_d=torch.from_numpy(z["in_sun"]).to(dev), xyz=torch.from_numpy(z["in_xyz"]).to(dev),
                z_vals=torch.from_numpy(z["in_z"]).to(dev)), torch.from_numpy(z["in_t"]).to(dev))
        for k in z.files:
            if k.startswith("out_") and k != "out_semantic_label":
                assert max_abs(r[k[4:]].cpu(), z[k]) <= OUT_TOL, (name, k)


def test_ragged_and_multi_chunk_sizes():
    """N*S not a multiple of the 128-row tile, S > 64 (two wavefront chunks per ray), S < 64."""
    dev = _dev()
    # (the two W = 512 sizes: the folded projections' partial sums of a last row tile that is mostly / partly beyond the points)
    for (N, S, W) in [(37, 96, 64), (5, 130, 32), (129, 7, 32), (1, 64, 32), (3, 130, 512), (37, 96, 512)]:
        cfg = O.OracleCfg(fc_units=W, n_samples=S)
        pn = O.init_params_numpy(cfg, 3)
        emb = torch.from_numpy(O.init_embedding_numpy(cfg, 3))
        b = O.batch_to_torch(O.synthetic_batch(N, S, seed=N + S))
        gp = _gpu_params(pn, dev, requires_grad=True)
        emb_g = emb.clone().to(dev).requires_grad_(True)
        hip = _hip_render(cfg, gp, emb_g, b, dev)
        po = O.to_torch(pn, requires_grad=True)
        emb_o = emb.clone().requires_grad_(True)
        ora = O.render_rays(po, emb_o, cfg, b["rays"], b["extras"], b["u"])
        hip.pop("_z_vals"); ora.pop("_z_vals")
        _compare_outputs(hip, ora, cfg)
        bg = {k: v.to(dev) for k, v in b.items()}
        O.total_loss(O.training_losses(hip, bg, cfg, 2)).backward()
        O.total_loss(O.training_losses(ora, b, cfg, 2)).backward()
        for k in po:
            err = rel_err(gp[k].grad.cpu(), po[k].grad)
            assert err <= GRAD_REL_TOL or max_abs(gp[k].grad.cpu(), po[k].grad) <= 1e-7 + GRAD_ABS_ESCAPE * float(po[k].grad.abs().max()), (N, S, k, err)
        assert rel_err(emb_g.grad.cpu(), emb_o.grad) <= GRAD_REL_TOL


REDUCED_STATS = []   # measured errors of the one-plane mode (printed by tests/conftest.py's summary hook when present)


def test_reduced_precision_mode_full_width(monkeypatch):
    """SNERF_FLAG_F16X1 -- one fp16 plane of the block-scaled tensors, one product: the mode of the reference's `precision = 16`
    runs (BASELINE.json configs[2] / [4]) -- at the headline width on the reference's own W = 512 fixture, through the SAME kernels
    as the default (templated on the plane count).  REDUCED precision.  The reference publishes no half-precision numbers and runs no
    such path here, so the yardsticks are made by the reference itself (tools/gen_golden.py: `yard_*_noise_*` of the fixture): the same
    case from weights carrying 1e-3 relative noise moves its outputs by 5.6e-3 abs, its gradients by 3.5 % relative L2 (3e-4 noise:
    1.7e-3, 1.0 %).  Bars: outputs within the 1e-3 yardstick (measured 7.4e-4: below even the 3e-4 one), loss terms 1 %, gradients
    within 3x the 1e-3 yardstick of the default arithmetic's (measured 3.8 %) and finite.  The default is held to 1e-4 / 2e-4 right above."""
    from snerf_amd import ops, _lib
    dev = _dev()
    z, meta, cfg = load_fixture("sem_siren_full")
    pn = fixture_params(z, meta, cfg)
    b = fixture_batch(z)
    grads, outs = {}, {}
    yard_out, yard_grad = float(z["yard_out_abs_noise_1e-3"]), float(z["yard_grad_rel_noise_1e-3"])
    assert 1e-3 < yard_out < 2e-2 and 1e-2 < yard_grad < 1e-1
    for name, flags, out_tol, loss_tol in (("f16x2", 0, OUT_TOL, 2e-4), ("f16x1", _lib.FLAG_F16X1, yard_out, 1e-2)):
        monkeypatch.setattr(ops, "BASE_FLAGS", flags)
        gp = _gpu_params(pn, dev, requires_grad=True)
        emb_g = torch.from_numpy(O.init_embedding_numpy(cfg, meta["seed"])).to(dev).requires_grad_(True)
        hip = _hip_render(cfg, gp, emb_g, b, dev)
        hip.pop("_z_vals")
        worst = 0.0
        for k in z.files:
            if k.startswith("out_") and k != "out_semantic_label_coarse":
                e = max_abs(hip[k[4:]].detach().cpu(), z[k])
                worst = max(worst, e)
                assert e <= out_tol, (name, k, e)
        bg = {k: v.to(dev) for k, v in b.items()}
        ld = O.training_losses(hip, bg, cfg, meta["epoch"])
        for k in ld:
            ref = float(z["loss_" + k])
            assert abs(float(ld[k].detach()) - ref) <= loss_tol * max(1.0, abs(ref)), (name, k, float(ld[k].detach()), ref)
        O.total_loss(ld).backward()
        grads[name] = {k: v.grad.clone().cpu() for k, v in gp.items()}
        outs[name] = worst
    assert all(torch.isfinite(g).all() for g in grads["f16x1"].values())
    rel = {k: float(rel_err(grads["f16x1"][k], grads["f16x2"][k])) for k in grads["f16x2"] if float(grads["f16x2"][k].abs().max()) > 0}
    REDUCED_STATS.append({"worst_output_abs_err": outs["f16x1"], "worst_grad_rel_l2": max(rel.values()), "default_worst_output_abs_err": outs["f16x2"],
                          "reference_under_1e-3_weight_noise": {"outputs_abs": yard_out, "grad_rel_l2": yard_grad},
                          "reference_under_3e-4_weight_noise": {"outputs_abs": float(z["yard_out_abs_noise_3e-4"]), "grad_rel_l2": float(z["yard_grad_rel_noise_3e-4"])}})
    print("f16x1 at W=512:", REDUCED_STATS[-1])
    assert max(rel.values()) <= 3.0 * yard_grad, sorted(rel.items(), key=lambda kv: -kv[1])[:3]


def test_forward_backward_capture_in_a_hip_graph():
    """include/snerf_hip.h promises hot calls that allocate nothing, never synchronise and read no environment: one pass
    (pack -> forward -> backward -> unpack, through ops.render_pass and autograd) is captured in a torch.cuda.CUDAGraph
    (hipStreamBeginCapture underneath), replayed twice, and must reproduce the eager call's outputs and gradients bit for bit."""
    from snerf_amd import ops
    dev = _dev()
    cfg = O.OracleCfg(fc_units=64, n_samples=16)
    pn = O.init_params_numpy(cfg, 3)
    b = O.batch_to_torch(O.synthetic_batch(256, 16, seed=5))
    emb = torch.from_numpy(O.init_embedding_numpy(cfg, 3)).to(dev)
    bg = {k: v.to(dev) for k, v in b.items()}

    from snerf_amd import ops
    spec = _spec(cfg)
    rays, extras, u = bg["rays"], bg["extras"], bg["u"]
    t = emb[extras[:, 3].long()]
    zs = torch.linspace(0, 1, cfg.n_samples).to(dev)

    def run(gp):       # device-resident inputs only: nothing below may copy from the host or synchronise
        for p in gp.values():
            p.grad = None
        packed = ops.pack_params(spec, gp)
        res = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, None, packed=packed)
        sc = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_vals=res["z_vals"]), t, None, sc_pass=True, packed=packed)
        loss = res["rgb"].square().sum() + res["depth"].sum() + res["semantic_logits"].sum() + sc["sun"].sum()
        loss.backward()
        outs = {k: v.detach().clone() for k, v in res.items() if v.is_floating_point()}
        outs["sun_sc"] = sc["sun"].detach().clone()
        return outs, {k: p.grad.clone() for k, p in gp.items() if p.grad is not None}

    gp = _gpu_params(pn, dev, requires_grad=True)
    out_e, grad_e = run(gp)                       # eager reference (also warms every lazy initialisation up)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                 # a warm-up on the capture stream, as torch asks for
        run(gp)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g, grad_g = run(gp)
    for _ in range(3):   # later replays too: memset nodes / large by-value arguments once made them differ from the first
        graph.replay()
        torch.cuda.synchronize()
        for k in out_e:
            assert torch.equal(out_g[k], out_e[k]), (k, max_abs(out_g[k].cpu(), out_e[k].cpu()))
        for k in grad_e:
            assert torch.equal(grad_g[k], grad_e[k]), (k, max_abs(grad_g[k].cpu(), grad_e[k].cpu()))
